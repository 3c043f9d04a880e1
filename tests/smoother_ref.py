"""TEST-ONLY helpers shared by tests/test_smoother.py, tests/test_oracle.py and tests/golden/make_golden.py: the RTS
backward recursion of mav_state_est.cpp:98-189 driven with the ORACLE's po_ekf_smoothing_step (rbis.cpp:234-266)."""
import ctypes as C

import numpy as np

from util import embed21


def start_of(w, zero_bias=False):
    """(vec, quat, P0, q4) of a run; zero_bias: a 21-state filter whose bias states are switched off (zero variance,
    zero process noise) -- the case the smoother's bias-block fix exists for (rbis.cpp:244-251)."""
    vec, quat, P0 = w.initial_state()
    q4 = w.process_noise()
    if zero_bias:
        P0[15:, :, :] = 0.0
        P0[:, 15:, :] = 0.0
        vec[15:] = 0.0
        q4 = np.array([q4[0], q4[1], 0.0, 0.0])
    return vec, quat, P0, q4


def oracle_forward(oracle, w, n, T, B, zero_bias=False):
    """Forward pass with the oracle, keeping (pred, filtered) posteriors of every step (INS update, then legodo)."""
    vec, quat, P0, q4 = start_of(w, zero_bias)
    v21, P21 = embed21(vec, P0)
    ob = oracle.OracleBatch(v21, quat, P21)
    hist = []
    for k in range(T):
        imu = w.imu_block(k)
        lo, mask = w.legodo_block(k)
        ob.predict(imu, q4)
        pred = (ob.vec.copy(), ob.quat.copy(), ob.cov.copy())
        ob.update_indexed([3, 4, 5], lo[0:3], lo[3:6], mask=mask)
        hist.append((pred, (ob.vec.copy(), ob.quat.copy(), ob.cov.copy())))
    return hist


def oracle_smooth_step(oracle, nxt_pred, nxt, cur, dt):
    L = oracle.lib()
    B = cur[0].shape[1]
    out_v, out_q, out_P = cur[0].copy(), cur[1].copy(), cur[2].copy()

    def mk(v, q, b):
        s = oracle.Rbis()
        s.vec[:] = list(v[:, b])
        s.quat[:] = list(q[:, b])
        return s

    def mkP(P, b):
        m = oracle.Rbim()
        m.m[:] = list(np.ascontiguousarray(P[:, :, b].T).ravel())
        return m
    for b in range(B):
        sp, Pp = mk(nxt_pred[0], nxt_pred[1], b), mkP(nxt_pred[2], b)
        sn, Pn = mk(nxt[0], nxt[1], b), mkP(nxt[2], b)
        sc, Pc = mk(cur[0], cur[1], b), mkP(cur[2], b)
        L.po_ekf_smoothing_step(C.byref(sp), C.byref(Pp), C.byref(sn), C.byref(Pn), dt, C.byref(sc), C.byref(Pc))
        out_v[:, b] = sc.vec[:]
        out_q[:, b] = sc.quat[:]
        out_P[:, :, b] = np.array(Pc.m[:]).reshape(21, 21).T
    return out_v, out_q, out_P


def oracle_backward_pass(oracle, w, n, T, B, dt, keep):
    """Forward pass + full backward recursion with the oracle; returns {k: (vec, quat, cov)} for the steps in `keep`."""
    hist = oracle_forward(oracle, w, n, T, B)
    nxt = hist[T - 1][1]
    out = {}
    for k in range(T - 2, -1, -1):
        nxt = oracle_smooth_step(oracle, hist[k + 1][0], nxt, hist[k][1], dt)
        if k in keep:
            out[k] = nxt
    return out

"""CPU tests of the oracle itself (PARITY UNPINNED: the reference has no vectors).

Pins: (1) C oracle == independent numpy restatement to <=1e-12; (2) analytic known-answer tests (i)-(vii) of
SURVEY.md 8c; (3) the committed golden trajectories.
"""
import os

import numpy as np
import pytest

from util import diag_full, embed21, pad_z, random_spd, rel, run_config

from oracle import numpy_restatement as nr
from pronto_amd.synth import Workload

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _ident_state(B):
    vec = np.zeros((21, B))
    quat = np.zeros((4, B))
    quat[0] = 1
    return vec, quat


class NumpyFilter:
    """numpy_restatement driven through the same calls as OracleBatch."""

    def __init__(self, vec, quat, cov):
        self.v, self.q, self.P = vec.T.copy(), quat.T.copy(), np.transpose(cov, (2, 0, 1)).copy()
        self.ll = np.zeros(vec.shape[1])
        self.q4 = None

    def predict(self, imu, q4):
        self.v, self.q, self.P = nr.predict(self.v, self.q, self.P, imu[0:3].T, imu[3:6].T, imu[6], q4)

    def update_indexed(self, idx, z, Rd, quat_meas=None, mask=None):
        v, q, P, ll = nr.update(self.v, self.q, self.P, self.ll, idx, z.T, diag_full(Rd),
                                quat_meas=None if quat_meas is None else quat_meas.T)
        m = np.ones(len(ll), bool) if mask is None else mask.astype(bool)
        self.v[m], self.q[m], self.P[m], self.ll[m] = v[m], q[m], P[m], ll[m]


@pytest.mark.parametrize("n,vo,sm", [(21, 32, 25), (15, 32, 0)])
def test_c_oracle_matches_numpy_restatement(oracle, n, vo, sm):
    B, T = 48, 150
    w = Workload(B, n_states=n)
    vec, quat, P0 = w.initial_state()
    P0 = P0 + random_spd(n, B, 0.03, 7)
    v21, P21 = embed21(vec, P0)
    ob = oracle.OracleBatch(v21, quat, P21)
    nf = NumpyFilter(v21, quat, P21)
    run_config(ob, w, T, vo_every=vo, sm_every=sm)
    run_config(nf, w, T, vo_every=vo, sm_every=sm)
    assert rel(ob.vec.T, nf.v) < 1e-12
    assert rel(ob.quat.T, nf.q) < 1e-12
    assert rel(np.transpose(ob.cov, (2, 0, 1)), nf.P) < 1e-12
    assert rel(ob.ll, nf.ll) < 1e-12


def test_kat_stationary(oracle):
    """(i) gyro = 0, accel = -R^T g_vec  =>  v and position constant for any dt."""
    g, _ = oracle.constants()
    B = 5
    rng = np.random.default_rng(0)
    vec, quat = _ident_state(B)
    q = rng.normal(size=(4, B))
    quat = q / np.linalg.norm(q, axis=0)
    pos0 = rng.normal(size=(3, B))
    vec[9:12] = pos0
    ob = oracle.OracleBatch(vec, quat, np.zeros((21, 21, B)))
    Rm = nr.rot_of_quat(quat.T)  # body -> world
    accel = np.einsum("bji,j->ib", Rm, np.array([0, 0, g]))
    for dt in (1e-3, 1e-2, 0.05):
        imu = np.zeros((7, B))
        imu[3:6] = accel
        imu[6] = dt
        ob.predict(imu, [1e-4, 1e-2, 0, 0])
    assert np.max(np.abs(ob.vec[3:6])) < 1e-14
    assert np.max(np.abs(ob.vec[9:12] - pos0)) < 1e-14


def test_kat_constant_yaw_rate(oracle):
    """(ii) constant omega about z, level: quat(t) = Exp(z omega t) exactly (Euler on SO(3) is exact for a fixed axis)."""
    g, _ = oracle.constants()
    B, wz, dt, N = 3, 0.7, 1e-3, 500
    vec, quat = _ident_state(B)
    ob = oracle.OracleBatch(vec, quat, np.zeros((21, 21, B)))
    imu = np.zeros((7, B))
    imu[2] = wz
    imu[5] = g
    imu[6] = dt
    for _ in range(N):
        ob.predict(imu, [0, 0, 0, 0])
    ang = wz * dt * N
    assert np.allclose(ob.quat[0], np.cos(ang / 2), atol=1e-13)
    assert np.allclose(ob.quat[3], np.sin(ang / 2), atol=1e-13)
    assert np.max(np.abs(ob.quat[1:3])) < 1e-15
    assert np.max(np.abs(ob.vec[6:9])) == 0.0  # chi folded every step (|omega dt| >> tol)


def test_kat_minimal_ac_covariance(oracle):
    """(iii) prior omega = v = 0, quat = I, diagonal P: P' = Ad P Ad^T + Qd from hand-computed blocks."""
    g, _ = oracle.constants()
    dt, qg, qa, qbg, qba = 0.01, 3e-4, 2e-2, 1e-6, 1e-5
    pv, pc, pp, pbg, pba = 0.04, 0.003, 0.25, 1e-4, 1e-3
    vec, quat = _ident_state(1)
    P = np.zeros((21, 21, 1))
    for lo, val in ((3, pv), (6, pc), (9, pp), (15, pbg), (18, pba)):
        for i in range(3):
            P[lo + i, lo + i, 0] = val
    ob = oracle.OracleBatch(vec, quat, P)
    imu = np.zeros((7, 1))
    imu[5] = g
    imu[6] = dt
    ob.predict(imu, [qg, qa, qbg, qba])
    C = ob.cov[:, :, 0]
    G = np.array([[0, g, 0], [-g, 0, 0], [0, 0, 0.0]])  # skew(R^T g_vec), g_vec = (0,0,-g)
    I3 = np.eye(3)
    exp = np.zeros((21, 21))
    exp[0:3, 0:3] = qg * I3
    exp[12:15, 12:15] = qa * I3
    exp[3:6, 3:6] = pv * I3 + dt * dt * (pc * G @ G.T + pba * I3) + qa * dt * I3
    exp[3:6, 6:9] = dt * G * pc
    exp[3:6, 9:12] = dt * pv * I3 + dt * dt * pc * G @ np.zeros((3, 3))  # [Delta,chi] block is -R vhat = 0
    exp[3:6, 18:21] = -dt * pba * I3
    exp[6:9, 6:9] = pc * I3 + dt * dt * pbg * I3 + qg * dt * I3
    exp[6:9, 15:18] = -dt * pbg * I3
    exp[9:12, 9:12] = pp * I3 + dt * dt * pv * I3
    exp[15:18, 15:18] = (pbg + qbg * dt) * I3
    exp[18:21, 18:21] = (pba + qba * dt) * I3
    exp = np.triu(exp) + np.triu(exp, 1).T
    # second-order cross term of [Delta, chi]: row Delta of Ad = [dt I (v), 0 (chi)], row chi = [I], so
    # P'[Delta,chi] = dt * P[v,chi] = 0; P'[Delta,v] includes dt*pv (above) -- nothing else.
    assert np.max(np.abs(C - exp)) < 1e-16 + 1e-13 * np.max(np.abs(exp))


def test_kat_scalar_update(oracle):
    """(iv) m = 1: K = P[:,i]/(P_ii+R), dP = K P[i,:], ll = -log(P_ii+R) - r^2/(P_ii+R)."""
    B, i, R = 4, 8, 0.02
    vec, quat = _ident_state(B)
    rng = np.random.default_rng(3)
    vec[:] = rng.normal(size=vec.shape) * 0.1
    vec[6:9] = 0
    P = random_spd(21, B, 0.2, 4)
    z = rng.normal(size=(1, B))
    ob = oracle.OracleBatch(vec, quat, P)
    ob.update_indexed([i], z, np.full((1, B), R))
    s = P[i, i] + R
    r = z[0] - vec[i]
    K = P[:, i] / s
    assert rel(ob.cov, P - K[:, None, :] * P[i][None, :, :]) < 1e-13
    assert rel(ob.ll, -np.log(s) - r * r / s) < 1e-13
    dx = K * r
    keep = [j for j in range(21) if j not in (6, 7, 8)]
    assert rel(ob.vec[keep], (vec + dx)[keep]) < 1e-13
    # the chi increment was folded into the quaternion
    assert np.max(np.abs(ob.vec[6:9])) == 0.0
    dq, _ = nr.quat_exp(dx[6:9].T)
    assert rel(ob.quat.T, dq) < 1e-13


def test_kat_orientation_update_zero_residual(oracle):
    """(v) q_meas = x.quat and z = x.pos  =>  residual 0: state unchanged, P reduced, ll = -log det S."""
    B = 3
    w = Workload(B, n_states=21)
    vec, quat, P0 = w.initial_state()
    P0 = P0 + random_spd(21, B, 0.02, 5)
    ob = oracle.OracleBatch(vec, quat, P0)
    idx = [9, 10, 11, 6, 7, 8]
    Rd = np.full((6, B), 1e-3)
    ob.update_indexed(idx, pad_z(vec[9:12], 6), Rd, quat_meas=quat)
    assert rel(ob.vec, vec) < 1e-15 and rel(ob.quat, quat) < 1e-15
    S = P0[np.ix_(idx, idx)] + np.eye(6)[:, :, None] * 1e-3
    logdet = np.array([np.linalg.slogdet(S[:, :, b])[1] for b in range(B)])
    assert rel(ob.ll, -logdet) < 1e-12
    assert np.all(np.diagonal(P0 - ob.cov, axis1=0, axis2=1) > -1e-15)


def test_kat_symmetry_psd_long_run(oracle):
    """(vi) P stays symmetric (to rounding) and PSD over 3000 steps although the reference never symmetrises."""
    B = 4
    w = Workload(B, n_states=21)
    vec, quat, P0 = w.initial_state()
    ob = oracle.OracleBatch(vec, quat, P0)
    run_config(ob, w, 3000, vo_every=32, sm_every=25)
    C = ob.cov
    assert np.max(np.abs(C - np.swapaxes(C, 0, 1))) < 1e-13 * np.max(np.abs(C))
    for b in range(B):
        ev = np.linalg.eigvalsh(0.5 * (C[:, :, b] + C[:, :, b].T))
        assert ev.min() > -1e-12
    assert np.all(np.isfinite(ob.vec)) and np.all(np.isfinite(ob.ll))


def test_kat_15_state_equals_21_state_zero_bias(oracle):
    """(vii) bias states and their covariance pinned to zero stay zero; entries 0-14 are the 15-state filter."""
    B = 8
    w = Workload(B, n_states=15)
    vec, quat, P0 = w.initial_state()
    v21, P21 = embed21(vec, P0)
    ob = oracle.OracleBatch(v21, quat, P21)
    run_config(ob, w, 200, vo_every=32)
    assert np.max(np.abs(ob.vec[15:])) == 0.0
    assert np.max(np.abs(ob.cov[15:])) == 0.0 and np.max(np.abs(ob.cov[:, 15:])) == 0.0


def test_handlers_arithmetic(oracle):
    """LegOdoCommon / getDeltaAsVelocity / fovis composition restatements (rbis_legodo_common.cpp:110-169)."""
    import ctypes as C
    L = oracle.lib()
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    r = np.array([0.2, 0.1, 0.3, 0.5, 0.9])
    pos = np.array([1.0, 2.0, 3.0])
    dt_ = np.array([0.002, -0.001, 0.0005])
    dq = np.array([1.0, 0, 0, 0])
    idx = (C.c_int * 6)()
    z, Rd = np.zeros(6), np.zeros(6)
    m = L.po_legodo_create_measurement(0, dp(r), dp(pos), dp(dt_), dp(dq), 2000, 1000, 1, C.c_float(0.0), idx, dp(z), dp(Rd))
    assert m == 3 and list(idx[:3]) == [3, 4, 5]
    assert np.allclose(z[:3], dt_ / 1e-3) and np.allclose(Rd[:3], 0.01)
    m = L.po_legodo_create_measurement(0, dp(r), dp(pos), dp(dt_), dp(dq), 2000, 1000, 1, C.c_float(1.0), idx, dp(z), dp(Rd))
    assert np.allclose(Rd[:3], 0.25)  # uncertain branch
    m = L.po_legodo_create_measurement(2, dp(r), dp(pos), dp(dt_), dp(dq), 2000, 1000, 0, C.c_float(0.0), idx, dp(z), dp(Rd))
    assert m == 3  # pos_and_lin_rate falls back to lin_rate when the position is not valid
    m = L.po_legodo_create_measurement(2, dp(r), dp(pos), dp(dt_), dp(dq), 2000, 1000, 1, C.c_float(0.0), idx, dp(z), dp(Rd))
    assert m == 6 and list(idx[:6]) == [9, 10, 11, 3, 4, 5] and np.allclose(z[:3], pos) and np.allclose(Rd[:3], 0.04)
    # euler <-> quat round trip
    q = np.zeros(4)
    rpy = np.zeros(3)
    L.po_euler_to_quat(0.1, -0.2, 0.3, dp(q))
    L.po_quat_to_euler(dp(q), dp(rpy))
    assert np.allclose(rpy, [0.1, -0.2, 0.3], atol=1e-15)
    # fovis composition = isometry product
    p0 = np.array([0.5, -1.0, 2.0])
    t = np.array([0.1, 0.2, 0.3])
    q1 = np.zeros(4)
    L.po_euler_to_quat(0.0, 0.0, 0.5, dp(q1))
    z3, qm = np.zeros(3), np.zeros(4)
    L.po_fovis_compose(dp(p0), dp(q), dp(t), dp(q1), dp(z3), dp(qm))
    Rq = nr.rot_of_quat(q[None])[0]
    assert np.allclose(z3, p0 + Rq @ t, atol=1e-15)
    assert np.allclose(qm, nr.quat_mul(q[None], q1[None])[0], atol=1e-15)


def test_smoothing_step_identity(oracle):
    """ekfSmoothingStep (rbis.cpp:234-266): if next == next_pred the smoothed state/cov are unchanged."""
    import ctypes as C
    L = oracle.lib()
    w = Workload(1, n_states=21)
    vec, quat, P0 = w.initial_state()
    cur, curP = oracle.Rbis(), oracle.Rbim()
    for i in range(21):
        cur.vec[i] = vec[i, 0]
    for i in range(4):
        cur.quat[i] = quat[i, 0]
    P = P0[:, :, 0] + 1e-3 * np.eye(21)
    for c in range(21):
        for r_ in range(21):
            curP.m[c * 21 + r_] = P[r_, c]
    nxt, nxtP, ll = oracle.Rbis(), oracle.Rbim(), C.c_double(0)
    g = (C.c_double * 3)(0.01, -0.02, 0.03)
    a = (C.c_double * 3)(0.1, 0.2, 9.7)
    L.po_imu_process_step(g, a, 0.01, 1e-4, 1e-2, 1e-8, 1e-8, C.byref(cur), C.byref(curP), 0.0, C.byref(nxt), C.byref(nxtP), C.byref(ll))
    before_v = np.array(cur.vec[:])
    before_P = np.array(curP.m[:])
    L.po_ekf_smoothing_step(C.byref(nxt), C.byref(nxtP), C.byref(nxt), C.byref(nxtP), 0.01, C.byref(cur), C.byref(curP))
    assert np.allclose(np.array(cur.vec[:]), before_v, atol=1e-12)
    assert np.allclose(np.array(curP.m[:]), before_P, atol=1e-12)


@pytest.mark.parametrize("name", ["n15_legodo", "n15_legodo_vo", "n21_legodo_scanmatch"])
def test_golden_trajectories(oracle, name):
    """The oracle reproduces the committed fixtures (tests/golden/make_golden.py) on this machine."""
    gold = np.load(os.path.join(GOLD, name + ".npz"))
    n, B, T, stride, vo, sm = (int(v) for v in gold["meta"][:6])
    assert (gold["meta"][6], gold["meta"][7]) == oracle.constants()
    w = Workload(B, n_states=n)
    vec, quat, P0 = w.initial_state()
    v21, P21 = embed21(vec, P0)
    ob = oracle.OracleBatch(v21, quat, P21)
    for s in range(T // stride):
        run_config(ob, w, stride, vo_every=vo, sm_every=sm, k0=s * stride)
        assert rel(ob.vec, gold["vec"][s]) < 1e-10
        assert rel(ob.quat, gold["quat"][s]) < 1e-10
        assert rel(np.stack([ob.cov[i, i] for i in range(21)]), gold["pdiag"][s]) < 1e-10
        assert rel(ob.ll, gold["ll"][s]) < 1e-10
    # the filter actually tracks the synthetic truth (sanity of the generator + oracle pair)
    tr = w.truth(w.time_s(T))
    assert np.abs(ob.vec[9:12] - tr["pos"]).max() < (0.25 if vo or sm else 1.5)
    assert np.abs(ob.vec[3:6] - tr["vel_b"]).max() < 0.3


@pytest.mark.parametrize("n", [15, 21])
def test_golden_smoother(oracle, n):
    """The oracle's backward recursion reproduces the committed smoother fixtures on this machine."""
    from smoother_ref import oracle_backward_pass
    gold = np.load(os.path.join(GOLD, "smoother_n%d.npz" % n))
    nn, B, T = (int(v) for v in gold["meta"][:3])
    assert nn == n and (gold["meta"][4], gold["meta"][5]) == oracle.constants()
    res = oracle_backward_pass(oracle, Workload(B, n_states=n), n, T, B, float(gold["meta"][3]), keep=(0, T // 2))
    for tag, k in (("first", 0), ("mid", T // 2)):
        for got, name in zip(res[k], ("vec_", "quat_", "cov_")):
            assert rel(got, gold[name + tag]) < 1e-10
    # smoothing shrinks the position uncertainty of the first step
    assert np.all(gold["cov_first"][9, 9] > 0)

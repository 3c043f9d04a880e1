"""Parity of the HIP path (through the C ABI) against the oracle -- run with `-m gpu` on an MI355X.

Tolerances (fp64; the HIP path uses the block structure of F and a rank-m downdate, the oracle is dense):
  per-run relative error <= 1e-9 on vec / quat / P / loglik blocks (observed ~1e-13), far inside the
  <=1e-5 pose/velocity bound of BASELINE.json.
"""
import os

import numpy as np
import pytest

from util import embed21, pad_z, random_spd, rel, rel_elem, run_config

from pronto_amd.synth import Workload

pytestmark = pytest.mark.gpu

TOL = 1e-9
ELEM_TOL = 1e-7   # element-wise bound with a floor of 1e-4 * max|block|: an entry 1e4 below the largest still gets 7 digits
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def pa():
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    from pronto_amd import _lib
    _lib.build()
    from pronto_amd import batch
    return batch


def make_pair(pa, oracle, w, dense_p0=None, n_snapshots=1):
    vec, quat, P0 = w.initial_state()
    if dense_p0:
        P0 = P0 + random_spd(w.n, w.B, 0.03, dense_p0)
    if w.n == 21:
        vec[15:18], vec[18:21] = 0.5 * w.bg, 0.5 * w.ba
    v21, P21 = embed21(vec, P0)
    ob = oracle.OracleBatch(v21, quat, P21)
    est = pa.BatchEstimator(w.B, n_states=w.n, device=0, n_snapshots=n_snapshots)
    est.set_constants(*oracle.constants())
    est.reset(vec, quat, P0)
    return est, ob


def check(est, ob, tol=TOL):
    n = est.n
    v, q, P, ll = est.get_head()
    errs = dict(vec=rel(v, ob.vec[:n]), quat=rel(q, ob.quat), cov=rel(P, ob.cov[:n, :n]), ll=rel(ll, ob.ll))
    assert max(errs.values()) < tol, errs
    # element-wise with an absolute floor (tests/util.py): small entries are held to 1e-4 * tol * max|block| absolute
    elem = dict(vec=rel_elem(v, ob.vec[:n]), quat=rel_elem(q, ob.quat), cov=rel_elem(P, ob.cov[:n, :n]),
                ll=rel_elem(ll, ob.ll))
    assert max(elem.values()) < ELEM_TOL, elem
    return errs


@pytest.mark.parametrize("n,B", [(15, 1), (15, 300), (21, 130)])
def test_fused_step_host_buffers(pa, oracle, n, B):
    """pb_step_legodo with PB_HOST blocks, ragged batch sizes (not a multiple of the wave), masks + uncertain R."""
    w = Workload(B, n_states=n)
    est, ob = make_pair(pa, oracle, w, dense_p0=3)
    q4 = w.process_noise()
    for k in range(120):
        imu = w.imu_block(k)
        lo, mask = w.legodo_block(k)
        est.step_legodo(imu, lo, mask, q4)
        ob.predict(imu, q4)
        ob.update_indexed([3, 4, 5], lo[0:3], lo[3:6], mask=mask)
    check(est, ob)


@pytest.mark.parametrize("generic", ["0", "1"])
@pytest.mark.parametrize("n", [15, 21])
def test_separate_calls_equal_fused(pa, oracle, n, generic, monkeypatch):
    """pb_predict + pb_update_indexed == pb_step_legodo == oracle, with the stand-alone update on the cooperative
    compile-time-index kernel (the handlers' index lists, default) and on the generic run-time-index kernel."""
    monkeypatch.setenv("PRONTO_BATCH_GENERIC_UPDATE", generic)
    B = 192
    w = Workload(B, n_states=n)
    est_f, ob = make_pair(pa, oracle, w)
    est_s, _ = make_pair(pa, oracle, w)
    q4 = w.process_noise()
    for k in range(60):
        imu = w.imu_block(k)
        lo, mask = w.legodo_block(k)
        est_f.step_legodo(imu, lo, mask, q4)
        est_s.predict(imu, q4)
        est_s.update_indexed([3, 4, 5], np.ascontiguousarray(lo[0:3]), np.ascontiguousarray(lo[3:6]), mask=mask)
        ob.predict(imu, q4)
        ob.update_indexed([3, 4, 5], lo[0:3], lo[3:6], mask=mask)
    check(est_f, ob)
    check(est_s, ob)
    vf, qf, Pf, lf = est_f.get_head()
    vs, qs, Ps, ls = est_s.get_head()
    assert rel(vf, vs) < 1e-12 and rel(Pf, Ps) < 1e-12 and rel(lf, ls) < 1e-12


@pytest.mark.parametrize("n", [15, 21])
@pytest.mark.parametrize("m", [1, 2, 3, 4, 5, 6])
def test_generic_update_every_m(pa, oracle, n, m):
    """k_update<NS,M,ORIENT>: random distinct index lists, the three R encodings, with and without orientation."""
    B = 100
    rng = np.random.default_rng(100 * n + m)
    w = Workload(B, n_states=n)
    est, ob = make_pair(pa, oracle, w, dense_p0=5)
    q4 = w.process_noise()
    for trial in range(4):
        imu = w.imu_block(trial)
        est.predict(imu, q4)
        ob.predict(imu, q4)
        idx = [int(i) for i in rng.choice(n, size=m, replace=False)]
        z = np.ascontiguousarray(ob.vec[idx] + 0.05 * rng.normal(size=(m, B)))
        mask = (rng.random(B) > 0.2).astype(np.uint8)
        orient = trial % 2 == 1
        qm = None
        if orient:
            d = 0.02 * rng.normal(size=(3, B))
            from oracle import numpy_restatement as nr
            qm = np.ascontiguousarray(nr.quat_mul(ob.quat.T, nr.quat_exp(d.T)[0]).T)
        kind = trial % 3
        if kind == 0:      # broadcast diagonal
            rb = list(0.01 + 0.05 * rng.random(m))
            Rd = np.tile(np.array(rb)[:, None], (1, B))
            est.update_indexed(idx, z, rb, mask=mask, quat_meas=qm)
            ob.update_indexed(idx, z, Rd, quat_meas=qm, mask=mask)
        elif kind == 1:    # per-filter diagonal
            Rd = np.ascontiguousarray(0.01 + 0.05 * rng.random((m, B)))
            est.update_indexed(idx, z, Rd, mask=mask, quat_meas=qm)
            ob.update_indexed(idx, z, Rd, quat_meas=qm, mask=mask)
        else:              # per-filter full R with a diagonal matrix in it (the oracle driver takes diagonals)
            Rd = 0.01 + 0.05 * rng.random((m, B))
            Rf = np.zeros((m * m, B))
            for i in range(m):
                Rf[i * m + i] = Rd[i]
            # (for m = 1 a [1,B] array is a diagonal and a full R at once)
            est.update_indexed(idx, z, np.ascontiguousarray(Rf), mask=mask, quat_meas=qm)
            ob.update_indexed(idx, z, Rd, quat_meas=qm, mask=mask)
    check(est, ob)


def test_full_r_offdiagonal(pa, oracle):
    """PB_R_FULL with a genuinely non-diagonal R against the oracle's single-filter entry point."""
    import ctypes as C
    B, n, m = 70, 15, 3
    rng = np.random.default_rng(9)
    w = Workload(B, n_states=n)
    est, ob = make_pair(pa, oracle, w, dense_p0=6)
    idx = [9, 4, 7]
    z = np.ascontiguousarray(ob.vec[idx] + 0.05 * rng.normal(size=(m, B)))
    A = rng.normal(size=(B, m, m)) * 0.1
    R = np.einsum("bij,bkj->bik", A, A) + 0.01 * np.eye(m)
    Rf = np.ascontiguousarray(np.transpose(R, (2, 1, 0)).reshape(m * m, B))  # [c*m+r, b]
    est.update_indexed(idx, z, Rf)
    L = oracle.lib()
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    exp_v, exp_P, exp_ll = np.zeros((21, B)), np.zeros((21, 21, B)), np.zeros(B)
    for b in range(B):
        x, Pm = oracle.Rbis(), oracle.Rbim()
        for i in range(21):
            x.vec[i] = ob.vec[i, b]
        for i in range(4):
            x.quat[i] = ob.quat[i, b]
        Pm.m[:] = list(np.ascontiguousarray(ob.cov[:, :, b].T).ravel())
        ll = C.c_double(0)
        zz = np.ascontiguousarray(z[:, b])
        Rb = np.ascontiguousarray(R[b].T).ravel()
        ia = (C.c_int * m)(*idx)
        L.po_indexed_update(m, ia, dp(zz), dp(Rb), C.byref(x), C.byref(Pm), 0.0, C.byref(x), C.byref(Pm), C.byref(ll))
        exp_v[:, b] = x.vec[:]
        exp_P[:, :, b] = np.array(Pm.m[:]).reshape(21, 21).T
        exp_ll[b] = ll.value
    v, q, P, ll = est.get_head()
    assert rel(v, exp_v[:n]) < TOL and rel(P, exp_P[:n, :n]) < TOL and rel(ll, exp_ll) < TOL


@pytest.mark.parametrize("n", [15, 21])
@pytest.mark.parametrize("idx,orient", [([3, 4, 5], False), ([9, 10, 11], False), ([9, 10, 11, 3, 4, 5], False),
                                        ([9, 10, 11, 6, 7, 8], True), ([9, 10, 11, 8], True),
                                        ([8, 9, 10, 11], False), ([6, 7, 8, 9, 10, 11], False), ([11], False),  # (the GPF's substates)
                                        ([3, 4, 5, 0, 1, 2], False)])  # LegOdoCommon lin_rot_rate
def test_full_r_on_the_handler_index_lists(pa, oracle, n, idx, orient):
    """PB_R_FULL (pronto::indexed_measurement_t carries a full R_effective; the laser GPF's is genuinely non-diagonal) on the
    handlers' own index lists: runs on the compile-time-index kernels like a diagonal R does, against the oracle's
    single-filter entry points; a mask skips a third of the filters."""
    import ctypes as C
    B, m = 173, len(idx)
    rng = np.random.default_rng(n + m)
    w = Workload(B, n_states=n)
    est, ob = make_pair(pa, oracle, w, dense_p0=6)
    q4 = w.process_noise()
    for k in range(5):
        lo, mk = w.legodo_block(k)
        est.step_legodo(w.imu_block(k), lo, mk, q4)
        ob.predict(w.imu_block(k), q4)
        ob.update_indexed([3, 4, 5], lo[0:3], lo[3:6], mask=mk)
    z = np.ascontiguousarray(ob.vec[idx] + 0.05 * rng.normal(size=(m, B)))
    dq = np.concatenate([np.ones((1, B)), 0.02 * rng.normal(size=(3, B))])
    dq /= np.linalg.norm(dq, axis=0)
    a, b_ = ob.quat, dq
    qm = np.ascontiguousarray(np.stack([a[0] * b_[0] - a[1] * b_[1] - a[2] * b_[2] - a[3] * b_[3],
                                        a[0] * b_[1] + a[1] * b_[0] + a[2] * b_[3] - a[3] * b_[2],
                                        a[0] * b_[2] + a[2] * b_[0] + a[3] * b_[1] - a[1] * b_[3],
                                        a[0] * b_[3] + a[3] * b_[0] + a[1] * b_[2] - a[2] * b_[1]]))
    A = rng.normal(size=(B, m, m)) * 0.1
    R = np.einsum("bij,bkj->bik", A, A) + 0.01 * np.eye(m)
    Rf = np.ascontiguousarray(np.transpose(R, (2, 1, 0)).reshape(m * m, B))  # [c*m+r, b]
    mask = (np.arange(B) % 3 != 1).astype(np.uint8)
    est.update_indexed(idx, z, Rf, quat_meas=qm if orient else None, mask=mask)
    L = oracle.lib()
    dp = lambda arr: arr.ctypes.data_as(C.POINTER(C.c_double))
    exp_v, exp_q, exp_P, exp_ll = ob.vec.copy(), ob.quat.copy(), ob.cov.copy(), ob.ll.copy()
    for b in range(B):
        if not mask[b]:
            continue
        x, Pm = oracle.Rbis(), oracle.Rbim()
        for i in range(21):
            x.vec[i] = ob.vec[i, b]
        for i in range(4):
            x.quat[i] = ob.quat[i, b]
        Pm.m[:] = list(np.ascontiguousarray(ob.cov[:, :, b].T).ravel())
        ll = C.c_double(0)
        zz = np.ascontiguousarray(z[:, b])
        Rb = np.ascontiguousarray(R[b].T).ravel()
        ia = (C.c_int * m)(*idx)
        if orient:
            qq = np.ascontiguousarray(qm[:, b])
            L.po_indexed_orient_update(m, ia, dp(zz), dp(Rb), dp(qq), C.byref(x), C.byref(Pm), float(ob.ll[b]), C.byref(x), C.byref(Pm), C.byref(ll))
        else:
            L.po_indexed_update(m, ia, dp(zz), dp(Rb), C.byref(x), C.byref(Pm), float(ob.ll[b]), C.byref(x), C.byref(Pm), C.byref(ll))
        exp_v[:, b] = x.vec[:]
        exp_q[:, b] = x.quat[:]
        exp_P[:, :, b] = np.array(Pm.m[:]).reshape(21, 21).T
        exp_ll[b] = ll.value
    v, q, P, ll = est.get_head()
    assert rel(v, exp_v[:n]) < TOL and rel(q, exp_q) < TOL and rel(P, exp_P[:n, :n]) < TOL and rel(ll, exp_ll) < TOL


def test_config3_vo_with_history_snapshot(pa, oracle):
    """BASELINE config 3 in miniature: predict + legodo every step, FovisHandler position_orient every 32nd step
    with T0 taken from the filter's own posterior at the previous VO time (pb_snapshot / pb_compose_delta)."""
    import ctypes as C
    import torch
    from oracle import numpy_restatement as nr
    B, n, T = 256, 15, 200
    w = Workload(B, n_states=n)
    est, ob = make_pair(pa, oracle, w, n_snapshots=2)
    q4 = w.process_noise()
    dev = torch.device("cuda:0")
    z_out = torch.empty((3, B), dtype=torch.float64, device=dev)
    q_out = torch.empty((4, B), dtype=torch.float64, device=dev)
    L = oracle.lib()
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    est.snapshot(1)
    snap_pos, snap_quat = ob.vec[9:12].copy(), ob.quat.copy()
    tr_prev = w.truth(w.time_s(0))
    for k in range(T):
        imu = w.imu_block(k)
        lo, mask = w.legodo_block(k)
        est.step_legodo(imu, lo, mask, q4)
        ob.predict(imu, q4)
        ob.update_indexed([3, 4, 5], lo[0:3], lo[3:6], mask=mask)
        if k % 32 == 31:
            # VO delta = truth motion between the two keyframes (+ noise via vo_block's pose)
            zt, qt, Rd = w.vo_block(k)
            Rp = nr.rot_of_quat(tr_prev["quat"].T)
            t_delta = np.ascontiguousarray(np.einsum("bji,jb->ib", Rp, zt - tr_prev["pos"]))
            qpc = tr_prev["quat"].T * np.array([1, -1, -1, -1.0])
            q_delta = np.ascontiguousarray(nr.quat_mul(qpc, qt.T).T)
            # HIP path: compose on the device, feed the device outputs straight into the m=6 update
            est.compose_delta(1, t_delta, q_delta, z_out, q_out)
            z6 = torch.zeros((6, B), dtype=torch.float64, device=dev)
            z6[0:3] = z_out
            est.update_indexed([9, 10, 11, 6, 7, 8], z6, torch.from_numpy(Rd).to(dev), quat_meas=q_out)
            # oracle: po_fovis_compose per filter, then the m=6 orientation update
            zc, qc = np.zeros((3, B)), np.zeros((4, B))
            for b in range(B):
                p0 = np.ascontiguousarray(snap_pos[:, b]); q0 = np.ascontiguousarray(snap_quat[:, b])
                tt = np.ascontiguousarray(t_delta[:, b]); qq = np.ascontiguousarray(q_delta[:, b])
                zo, qo = np.zeros(3), np.zeros(4)
                L.po_fovis_compose(dp(p0), dp(q0), dp(tt), dp(qq), dp(zo), dp(qo))
                zc[:, b], qc[:, b] = zo, qo
            assert rel(z_out.cpu().numpy(), zc) < 1e-12 and rel(q_out.cpu().numpy(), qc) < 1e-12
            ob.update_indexed([9, 10, 11, 6, 7, 8], pad_z(zc, 6), Rd, quat_meas=qc)
            # new keyframe
            est.snapshot(1)
            snap_pos, snap_quat = ob.vec[9:12].copy(), ob.quat.copy()
            tr_prev = w.truth(w.time_s(k + 1))
    check(est, ob)
    tr = w.truth(w.time_s(T))
    assert np.abs(ob.vec[9:12] - tr["pos"]).max() < 0.5


def test_config5_n21_scanmatch(pa, oracle):
    """BASELINE config 5 in miniature: 21-state filter, legodo every step, scan-match position_yaw (m=4) every 25th."""
    B, n, T = 128, 21, 150
    w = Workload(B, n_states=n)
    est, ob = make_pair(pa, oracle, w)
    run_config(est, w, T, sm_every=25)
    run_config(ob, w, T, sm_every=25)
    check(est, ob)


@pytest.mark.parametrize("name", ["n15_legodo", "n15_legodo_vo", "n21_legodo_scanmatch"])
def test_against_golden_fixtures(pa, oracle, name):
    """HIP path vs the committed golden trajectories (no oracle call: fixtures only)."""
    gold = np.load(os.path.join(GOLD, name + ".npz"))
    n, B, T, stride, vo, sm = (int(v) for v in gold["meta"][:6])
    w = Workload(B, n_states=n)
    vec, quat, P0 = w.initial_state()
    est = pa.BatchEstimator(B, n_states=n)
    est.set_constants(float(gold["meta"][6]), float(gold["meta"][7]))
    est.reset(vec, quat, P0)
    for s in range(T // stride):
        run_config(est, w, stride, vo_every=vo, sm_every=sm, k0=s * stride)
        v, q, P, ll = est.get_head()
        assert rel(v, gold["vec"][s][:n]) < TOL
        assert rel(q, gold["quat"][s]) < TOL
        assert rel(np.stack([P[i, i] for i in range(n)]), gold["pdiag"][s][:n]) < TOL
        assert rel(P[3:6, 6:12], gold["pvv"][s]) < TOL
        assert rel(ll, gold["ll"][s]) < TOL
    # BASELINE.json's bound: pose / velocity within 1e-5 relative of the reference CPU filter
    assert rel(v[9:12], gold["vec"][-1][9:12]) < 1e-5 and rel(v[3:6], gold["vec"][-1][3:6]) < 1e-5


def test_long_run_10k_steps(pa, oracle):
    """16 filters x 10 000 steps through pb_run_legodo against the golden long run."""
    import torch
    gold = np.load(os.path.join(GOLD, "n15_long.npz"))
    n, B, T, stride = (int(v) for v in gold["meta"][:4])
    w = Workload(B, n_states=n)
    vec, quat, P0 = w.initial_state()
    est = pa.BatchEstimator(B, n_states=n)
    est.set_constants(float(gold["meta"][6]), float(gold["meta"][7]))
    est.reset(vec, quat, P0)
    dev = torch.device("cuda:0")
    q4 = w.process_noise()
    for s in range(T // stride):
        imu, lo, mask = w.streams(s * stride, stride)
        est.run_legodo(torch.from_numpy(imu).to(dev), torch.from_numpy(lo).to(dev), torch.from_numpy(mask).to(dev), q4)
        v, q, P, ll = est.get_head()
        assert rel(v, gold["vec"][s][:n]) < 1e-8 and rel(q, gold["quat"][s]) < 1e-8
        assert rel(np.stack([P[i, i] for i in range(n)]), gold["pdiag"][s][:n]) < 1e-8
        assert rel(ll, gold["ll"][s]) < 1e-8


def test_full_size_64k_sampled_parity_and_properties(pa, oracle):
    """BASELINE config 2 at full size (65 536 filters): device-resident streams, one launch per step.
    Size-independent properties: (a) every filter's result is independent of the batch around it (a sampled
    sub-batch replayed alone gives the same bits), (b) sampled filters match the oracle, (c) everything finite,
    quaternions unit, (d) replays are bit-reproducible (checksum of checksums)."""
    import torch
    B, n, T = 65536, 15, 40
    w = Workload(B, n_states=n)
    vec, quat, P0 = w.initial_state()
    q4 = w.process_noise()
    imu, lo, mask = w.streams(0, T)
    dev = torch.device("cuda:0")
    d_imu, d_lo, d_mask = (torch.from_numpy(a).to(dev) for a in (imu, lo, mask))
    sums = []
    for rep in range(2):
        est = pa.BatchEstimator(B, n_states=n)
        est.set_constants(*oracle.constants())
        est.reset(vec, quat, P0)
        est.run_legodo(d_imu, d_lo, d_mask, q4)
        sums.append(est.summary())
        if rep == 0:
            v, q, P, ll = est.get_head()
        est.close()
    assert np.array_equal(sums[0], sums[1])                       # (d)
    assert sums[0][3] == 0 and sums[0][2] < 1e-12                  # (c) finite, |q|^2 = 1
    assert np.all(np.isfinite(v)) and np.all(np.isfinite(P)) and np.all(np.isfinite(ll))
    assert np.isclose(sums[0][0], ll.sum(), rtol=1e-12)
    sel = np.concatenate([np.arange(0, 64), np.arange(32700, 32764), np.arange(B - 128, B)])
    # (b) oracle on the sampled filters
    v21, P21 = embed21(vec[:, sel], P0[:, :, sel])
    ob = oracle.OracleBatch(v21, quat[:, sel], P21)
    ob.run_legodo(np.ascontiguousarray(imu[:, :, sel]), np.ascontiguousarray(lo[:, :, sel]),
                  np.ascontiguousarray(mask[:, sel]), q4)
    assert rel(v[:, sel], ob.vec[:n]) < TOL and rel(q[:, sel], ob.quat) < TOL
    assert rel(P[:, :, sel], ob.cov[:n, :n]) < TOL and rel(ll[sel], ob.ll) < TOL
    # (a) the same filters alone in a small batch: bit-identical
    est = pa.BatchEstimator(len(sel), n_states=n)
    est.set_constants(*oracle.constants())
    est.reset(np.ascontiguousarray(vec[:, sel]), np.ascontiguousarray(quat[:, sel]), np.ascontiguousarray(P0[:, :, sel]))
    est.run_legodo(torch.from_numpy(np.ascontiguousarray(imu[:, :, sel])).to(dev),
                   torch.from_numpy(np.ascontiguousarray(lo[:, :, sel])).to(dev),
                   torch.from_numpy(np.ascontiguousarray(mask[:, sel])).to(dev), q4)
    v2, q2, P2, ll2 = est.get_head()
    assert np.array_equal(v2, v[:, sel]) and np.array_equal(P2, P[:, :, sel]) and np.array_equal(ll2, ll[sel])


def test_kats_through_the_abi(pa, oracle):
    """Stationary and constant-yaw-rate KATs (SURVEY.md 8c i, ii) on the HIP path itself."""
    g, tol = oracle.constants()
    B = 64
    est = pa.BatchEstimator(B, n_states=15)
    est.set_constants(g, tol)
    quat = np.zeros(4); quat[0] = 1
    est.reset(np.zeros(15), quat, np.zeros((15, 15)), broadcast=True)
    imu = np.zeros((7, B))
    imu[2] = 0.7
    imu[5] = g
    imu[6] = 1e-3
    for _ in range(500):
        est.predict(imu, [0, 0, 0, 0])
    v, q, P, ll = est.get_head()
    ang = 0.7 * 1e-3 * 500
    assert np.allclose(q[0], np.cos(ang / 2), atol=1e-13) and np.allclose(q[3], np.sin(ang / 2), atol=1e-13)
    assert np.max(np.abs(v[3:6])) < 1e-13 and np.max(np.abs(v[9:12])) < 1e-13
    assert np.max(np.abs(P)) == 0.0 and np.all(ll == 0)


def test_error_behaviour_and_wire_layout(pa, oracle):
    """Error codes instead of exit()/exceptions in the library; rbisCreateFilterStateMessageCPP layout."""
    B = 10
    est = pa.BatchEstimator(B, n_states=15)
    w = Workload(B, n_states=15)
    imu = w.imu_block(0)
    with pytest.raises(pa.PbError) as e:
        est.predict(imu, [0, 0, 0, 0])          # before reset
    assert e.value.code == 4
    vec, quat, P0 = w.initial_state()
    est.reset(vec, quat, P0 + random_spd(15, B, 0.02, 1))
    z = np.zeros((1, B))
    for bad in ([15], [-1]):
        with pytest.raises(pa.PbError) as e:
            est.update_indexed(bad, z, [0.1])
        assert e.value.code == 1
    with pytest.raises(pa.PbError):
        est.update_indexed([3, 3], np.zeros((2, B)), [0.1, 0.1])    # duplicate index
    with pytest.raises(pa.PbError):
        est.update_indexed(list(range(7)), np.zeros((7, B)), [0.1] * 7)  # m > 6
    with pytest.raises(pa.PbError) as e:
        est.snapshot(5)
    assert e.value.code == 4
    with pytest.raises(pa.PbError) as e:
        pa.BatchEstimator(15_000_000, n_states=15)     # the [36][B] input blocks of one context must stay below 4 GiB
    assert e.value.code == 1 and "too large" in str(e.value)
    for call in (lambda: est.state_save(0), lambda: est.smooth_step(0, 1, 2, 3, 1e-3)):
        with pytest.raises(pa.PbError) as e:
            call()                                      # no checkpoint slots reserved
        assert e.value.code == 4
    v, q, P, ll = est.get_head()
    qq, s21, c21 = est.filter_state(3)
    assert np.array_equal(qq, q[:, 3]) and np.array_equal(s21[:15], v[:, 3]) and np.all(s21[15:] == 0)
    assert np.array_equal(c21[:15, :15], P[:, :, 3]) and np.all(c21[15:] == 0)
    # empty range and partial range queries
    v2, q2, P2, l2 = est.get_head(first=4, count=3)
    assert np.array_equal(v2, v[:, 4:7]) and np.array_equal(P2, P[:, :, 4:7])


def test_torch_stream_interop(pa, oracle):
    """Kernels launched on torch's current stream read torch-allocated HBM in place (no staging copy)."""
    import torch
    B, n = 512, 15
    w = Workload(B, n_states=n)
    est, ob = make_pair(pa, oracle, w)
    q4 = w.process_noise()
    dev = torch.device("cuda:0")
    side = torch.cuda.Stream()
    for k in range(12):
        imu = w.imu_block(k)
        lo, mask = w.legodo_block(k)
        if k % 2:   # a non-default torch stream: inputs are PRODUCED by torch kernels on it, the step follows in stream order
            with torch.cuda.stream(side):
                est.set_stream(side.cuda_stream)
                d_imu = torch.from_numpy(imu).to(dev, non_blocking=True) * 1.0
                d_lo = torch.from_numpy(lo).to(dev, non_blocking=True) + 0.0
                d_mask = torch.from_numpy(mask).to(dev, non_blocking=True).clone()
                est.step_legodo(d_imu, d_lo, d_mask, q4)
            side.synchronize()
        else:       # the default (null) stream, handle 0
            est.set_stream(torch.cuda.current_stream().cuda_stream)
            est.step_legodo(torch.from_numpy(imu).to(dev) * 1.0, torch.from_numpy(lo).to(dev), torch.from_numpy(mask).to(dev), q4)
            torch.cuda.current_stream().synchronize()
        ob.predict(imu, q4)
        ob.update_indexed([3, 4, 5], lo[0:3], lo[3:6], mask=mask)
    est.use_own_stream()
    torch.cuda.synchronize()
    check(est, ob)


@pytest.mark.parametrize("n,onelane", [(15, "0"), (15, "1"), (21, "0")])
@pytest.mark.parametrize("T_fuse", [1, 7, 64])
def test_time_fused_replay_equals_per_step_path(pa, oracle, T_fuse, n, onelane, monkeypatch):
    """pb_replay_legodo_fused (state resident in registers for T steps: the cooperative two-role kernel for 15 and 21 states,
    and the first one-lane 15-state kernel behind PRONTO_BATCH_REPLAY_ONELANE=1) gives the per-message path's results and
    matches the oracle; ragged last launch (50 steps in chunks of 7), ragged batch, masks, uncertain R."""
    import torch
    monkeypatch.setenv("PRONTO_BATCH_REPLAY_ONELANE", onelane)
    B, T = 1000, 50
    w = Workload(B, n_states=n)
    imu, lo, mask = w.streams(0, T)
    dev = torch.device("cuda:0")
    d = [torch.from_numpy(a).to(dev) for a in (imu, lo, mask)]
    q4 = w.process_noise()
    est_f, ob = make_pair(pa, oracle, w)
    est_s, _ = make_pair(pa, oracle, w)
    est_f.replay_legodo_fused(d[0], d[1], d[2], q4, T_fuse)
    est_s.run_legodo(d[0], d[1], d[2], q4)
    ob.run_legodo(imu, lo, mask, q4)
    check(est_f, ob)
    vf, qf, Pf, lf = est_f.get_head()
    vs, qs, Ps, ls = est_s.get_head()
    assert rel(vf, vs) < 1e-12 and rel(qf, qs) < 1e-12 and rel(Pf, Ps) < 1e-12 and rel(lf, ls) < 1e-12
    # (not bit for bit, although the cooperative replay runs the source of the per-step kernel's role bodies: the compiler
    # contracts multiply-adds differently in the two kernels)
    est_new = pa.BatchEstimator(8, n_states=21)
    with pytest.raises(pa.PbError):
        est_new.replay_legodo_fused(d[0][:, :, :8].contiguous(), d[1][:, :, :8].contiguous(), d[2][:, :8].contiguous(), q4, 4)   # before reset


@pytest.mark.parametrize("n", [15, 21])
def test_checkpointed_replay_keeps_every_posterior(pa, oracle, n):
    """pb_replay_legodo_checkpointed: the time-fused replay as a forward pass that keeps every posterior (history, smoother forward
    pass; mav_state_est.cpp:50-70,98-189).  Slot t must hold, BIT FOR BIT, what the same replay kernel leaves after t + 1 steps
    when it is launched one step at a time (the state's trip through memory is exact); it equals the per-message path with
    pb_set_output_slot to rounding (another kernel: other multiply-add contractions) and the oracle's trajectory; ragged batch,
    ragged last launch, a second call appending behind the first."""
    import torch
    B, T = 333, 23
    w = Workload(B, n_states=n)
    imu, lo, mask = w.streams(0, T)
    dev = torch.device("cuda:0")
    d = [torch.from_numpy(a).to(dev) for a in (imu, lo, mask)]
    q4 = w.process_noise()
    est_c, ob = make_pair(pa, oracle, w)      # checkpointed replay
    est_1, _ = make_pair(pa, oracle, w)       # the same kernel, one step per launch
    est_s, _ = make_pair(pa, oracle, w)       # per-message path into slots
    for e in (est_c, est_s):
        e.history_reserve(T + 2)
    with pytest.raises(pa.PbError):
        est_c.replay_legodo_checkpointed(d[0], d[1], d[2], q4, 7, first_slot=3)        # would run past the last slot
    est_c.replay_legodo_checkpointed(d[0][:16].contiguous(), d[1][:16].contiguous(), d[2][:16].contiguous(), q4, 7, first_slot=1)
    est_c.replay_legodo_checkpointed(d[0][16:].contiguous(), d[1][16:].contiguous(), d[2][16:].contiguous(), q4, 5, first_slot=17)
    head_c = est_c.get_head()
    for t in range(T):
        est_1.replay_legodo_fused(d[0][t:t + 1].contiguous(), d[1][t:t + 1].contiguous(), d[2][t:t + 1].contiguous(), q4, 1)
        est_s.set_output_slot(1 + t)
        est_s.step_legodo(d[0][t], d[1][t], d[2][t], q4)
        ob.predict(imu[t], q4)
        ob.update_indexed([3, 4, 5], np.ascontiguousarray(lo[t][0:3]), np.ascontiguousarray(lo[t][3:6]), mask=mask[t])
        want = est_1.get_head()
        est_c.state_restore(1 + t)
        got = est_c.get_head()
        for x, y in zip(got, want):
            assert np.array_equal(x, y), t                      # bit for bit
        est_s.state_restore(1 + t)
        for x, y in zip(got, est_s.get_head()):
            assert rel(x, y) < 1e-12, t
        if t in (0, T // 2, T - 1):
            check(est_c, ob)
    for x, y in zip(head_c, est_1.get_head()):                  # the head after the call = the last slot's content
        assert np.array_equal(x, y)
    for e in (est_c, est_1, est_s):
        e.close()


@pytest.mark.parametrize("generic", ["0", "1"])
@pytest.mark.parametrize("n,vo,sm", [(15, 32, 0), (21, 0, 25)])
def test_full_size_configs_3_and_5_sampled_parity(pa, oracle, n, vo, sm, generic, monkeypatch):
    """BASELINE configs 3 (n=15, +VO m=6 every 32nd step) and 5 (n=21, +scan-match m=4 every 25th) at the full
    65 536 filters with device-resident inputs; 192 sampled filters against the oracle, everything finite."""
    import torch
    monkeypatch.setenv("PRONTO_BATCH_GENERIC_UPDATE", generic)   # corrections on the cooperative / the generic update kernel
    B, T = 65536, 64
    w = Workload(B, n_states=n)
    vec, quat, P0 = w.initial_state()
    q4 = w.process_noise()
    dev = torch.device("cuda:0")
    est = pa.BatchEstimator(B, n_states=n)
    est.set_constants(*oracle.constants())
    est.reset(vec, quat, P0)
    sel = np.concatenate([np.arange(0, 64), np.arange(30000, 30064), np.arange(B - 64, B)])
    v21, P21 = embed21(vec[:, sel], P0[:, :, sel])
    ob = oracle.OracleBatch(v21, quat[:, sel], P21)
    up = lambda a, dt=None: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    for k in range(T):
        imu = w.imu_block(k)
        lo, mask = w.legodo_block(k)
        est.step_legodo(up(imu), up(lo), up(mask), q4)
        ob.predict(np.ascontiguousarray(imu[:, sel]), q4)
        ob.update_indexed([3, 4, 5], np.ascontiguousarray(lo[0:3, sel]), np.ascontiguousarray(lo[3:6, sel]),
                          mask=np.ascontiguousarray(mask[sel]))
        if vo and k % vo == vo - 1:
            z, qm, Rd = w.vo_block(k)
            est.update_indexed([9, 10, 11, 6, 7, 8], up(pad_z(z, 6)), up(Rd), quat_meas=up(qm))
            ob.update_indexed([9, 10, 11, 6, 7, 8], pad_z(z[:, sel], 6), np.ascontiguousarray(Rd[:, sel]),
                              quat_meas=np.ascontiguousarray(qm[:, sel]))
        if sm and k % sm == sm - 1:
            z, qm, Rd = w.scanmatch_block(k)
            est.update_indexed([9, 10, 11, 8], up(pad_z(z, 4)), up(Rd), quat_meas=up(qm))
            ob.update_indexed([9, 10, 11, 8], pad_z(z[:, sel], 4), np.ascontiguousarray(Rd[:, sel]),
                              quat_meas=np.ascontiguousarray(qm[:, sel]))
    s = est.summary()
    assert s[3] == 0 and s[2] < 1e-12
    v, q, P, ll = est.get_head()
    assert rel(v[:, sel], ob.vec[:n]) < TOL and rel(q[:, sel], ob.quat) < TOL
    assert rel(P[:, :, sel], ob.cov[:n, :n]) < TOL and rel(ll[sel], ob.ll) < TOL


@pytest.mark.parametrize("n,B,kern", [(15, 1000, "coop"), (15, 715, "coop"), (15, 715, "lane"), (21, 700, "coop"),
                                      (21, 700, "quad"), (21, 1000, "quad")])
def test_every_kernel_variant_is_bit_identical(pa, oracle, n, B, kern, monkeypatch):
    """pb_create picks the step kernel (one-lane / two-wave / four-wave), the workgroup order (XCD-contiguous or not) and the cache
    policy of the state round trip (default / sc1 stores / non-temporal) from the batch size, so the small batches of the
    other tests never reach most variants.  Force every combination on ragged grids (16, 12 and 11 workgroups: remainder
    0, 4 and 3 over the 8 XCDs): each step kernel is checked against the oracle, and workgroup order and cache policy
    must not change a single bit."""
    w = Workload(B, n_states=n)
    q4 = w.process_noise()
    vo, sm = (7, 0) if n == 15 else (0, 7)
    monkeypatch.setenv("PRONTO_BATCH_COOP15", "1" if kern == "coop" else "0")
    monkeypatch.setenv("PRONTO_BATCH_QUAD21", "1" if kern == "quad" else "0")
    ref = None
    for xcd in ("0", "1"):
        for hint in ("0", "1", "2"):
            monkeypatch.setenv("PRONTO_BATCH_XCD", xcd)
            monkeypatch.setenv("PRONTO_BATCH_MEMHINT", hint)
            est, ob = make_pair(pa, oracle, w, dense_p0=5)
            assert [t for t in ("coop", "quad") if t in est.hot_kernel()] == ([] if kern == "lane" else [kern])
            for k in range(30):  # fused step kernel
                lo, mask = w.legodo_block(k)
                est.step_legodo(w.imu_block(k), lo, mask, q4)
            run_config(est, w, 30, vo_every=vo, sm_every=sm, k0=30)  # predict-only kernel + k_update m = 3, 4 / 6
            head = est.get_head()
            if ref is None:
                for k in range(30):
                    lo, mask = w.legodo_block(k)
                    ob.predict(w.imu_block(k), q4)
                    ob.update_indexed([3, 4, 5], lo[0:3], lo[3:6], mask=mask)
                run_config(ob, w, 30, vo_every=vo, sm_every=sm, k0=30)
                check(est, ob)
                ref = head
            else:
                for a, b in zip(head, ref):
                    assert np.array_equal(a, b), (xcd, hint)
            est.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n,kern", [(15, "coop"), (15, "lane"), (21, "quad"), (21, "coop")])
def test_cache_blocked_replay_is_bit_identical_to_the_step_by_step_order(pa, oracle, n, kern, monkeypatch):
    """pb_run_legodo beyond the memory-side cache runs filter range outer / time inner over blocks of whole tiles (pb_create;
    forced here on 1 024 filters in four blocks of 256 with PRONTO_BATCH_BLOCKED / _BLOCK_FILTERS).  The filters are independent:
    with the same step kernel the blocked order must give the step-by-step order's head BIT FOR BIT, whatever the cache policy, and
    both are the oracle's."""
    import torch
    B, T = 1024, 12
    dev = torch.device("cuda:0")
    w = Workload(B, n_states=n)
    q4 = w.process_noise()
    imu, lo, mask = w.streams(0, T)
    d = [torch.from_numpy(a).to(dev) for a in (imu, lo, mask)]
    monkeypatch.setenv("PRONTO_BATCH_COOP15", "1" if kern == "coop" else "0")
    monkeypatch.setenv("PRONTO_BATCH_QUAD21", "1" if kern == "quad" else "0")
    heads = {}
    for blocked in ("0", "1"):
        for hint in (("0", "1", "2") if blocked == "1" else ("0",)):
            monkeypatch.setenv("PRONTO_BATCH_BLOCKED", blocked)
            monkeypatch.setenv("PRONTO_BATCH_BLOCK_FILTERS", "320")
            monkeypatch.setenv("PRONTO_BATCH_MEMHINT", hint)
            est, ob = make_pair(pa, oracle, w, dense_p0=5)
            assert est.run_block() == (256 if blocked == "1" else 0)
            est.run_legodo(*d, q4)
            est.run_legodo(*d, q4)          # (twice: the second call starts from a head the blocked order left)
            heads[(blocked, hint)] = est.get_head()
            if blocked == "0":
                for _ in range(2):
                    ob.run_legodo(imu, lo, mask, q4)
                check(est, ob)
            est.close()
    ref = heads[("0", "0")]
    for key, head in heads.items():
        for a, b in zip(head, ref):
            assert np.array_equal(a, b), key


@pytest.mark.parametrize("n,default_kernels", [(15, True), (21, True), (15, False), (21, False)])
def test_broadcast_inputs_equal_replicated_blocks(pa, oracle, n, default_kernels, monkeypatch):
    """PB_HOST_BROADCAST: one message for every filter of the batch (a parameter sweep replaying one robot's log) passed
    as [rows] host values must give bit-identical results to the same values replicated into [rows, B] blocks; a mask
    cannot be broadcast.  Also on the A/B kernels (one-lane 15-state step, two-wave 21-state step), where the m = 6
    correction has no kernel taking its measurement as arguments and the broadcast blocks must be staged instead."""
    import torch
    if not default_kernels:
        monkeypatch.setenv("PRONTO_BATCH_COOP15", "0")
        monkeypatch.setenv("PRONTO_BATCH_QUAD21", "0")
    B = 333
    w = Workload(B, n_states=n)
    vec, quat, P0 = w.initial_state()
    P0 = P0 + random_spd(n, B, 0.03, 11)
    ests = []
    for _ in range(2):
        e = pa.BatchEstimator(B, n_states=n, n_snapshots=1)
        e.set_constants(*oracle.constants())
        e.reset(vec, quat, P0)
        e.snapshot(0)
        ests.append(e)
    rep, bc = ests
    rng = np.random.default_rng(5)
    q4 = w.process_noise()
    dev = torch.device("cuda:0")
    tile = lambda a: np.ascontiguousarray(np.repeat(np.asarray(a, dtype=np.float64)[:, None], B, axis=1))
    for k in range(40):
        imu = np.concatenate([0.3 * rng.standard_normal(3), [0.1, -0.2, 9.8] + 0.2 * rng.standard_normal(3), [1e-3]])
        lo = np.concatenate([0.2 * rng.standard_normal(3), [0.01, 0.02, 0.015]])
        if k % 2:
            rep.step_legodo(tile(imu), tile(lo), None, q4)
            bc.step_legodo(imu, lo, None, q4)
        else:
            rep.predict(tile(imu), q4)
            bc.predict(imu, q4)
        if k % 5 == 4:   # m = 3 with one full R for everybody
            z = 0.2 * rng.standard_normal(3)
            A = rng.standard_normal((3, 3))
            R = np.ascontiguousarray((A @ A.T * 0.01 + 0.01 * np.eye(3)).T.ravel())
            rep.update_indexed([3, 4, 5], tile(z), tile(R))
            bc.update_indexed([3, 4, 5], z, R)
        if k % 8 == 7:   # VO: compose on the device from a broadcast delta, then the m = 6 orientation update
            t = 0.02 * rng.standard_normal(3)
            dq = np.array([1.0, 0.01, -0.02, 0.015]); dq /= np.linalg.norm(dq)
            outs = []
            for e, tt, qq in ((rep, tile(t), tile(dq)), (bc, t, dq)):
                zo = torch.zeros((6, B), dtype=torch.float64, device=dev)
                qo = torch.empty((4, B), dtype=torch.float64, device=dev)
                e.compose_delta(0, tt, qq, zo[0:3], qo)
                e.update_indexed([9, 10, 11, 6, 7, 8], zo, [4e-4] * 3 + [1e-4] * 3, quat_meas=qo)
                e.snapshot(0)
                outs.append((zo.cpu().numpy(), qo.cpu().numpy()))
            assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
        if k % 6 == 5:   # IMU + leg odometry + scan-match / VO correction in one call, every block broadcast
            from pronto_amd import _lib
            kind, m = (_lib.PB_CORR_POS_YAW, 4) if k % 12 == 5 else (_lib.PB_CORR_POS_ORIENT, 6)
            zc = 0.1 * rng.standard_normal(m)
            qc = np.array([0.995, 0.01, -0.02, 0.09]); qc /= np.linalg.norm(qc)
            Rc = [0.0025] * 3 + [3e-4] * (m - 3)
            rep.step_legodo_correct(tile(imu), tile(lo), None, q4, kind, tile(zc), Rc, tile(qc))
            bc.step_legodo_correct(imu, lo, None, q4, kind, zc, Rc, qc)
        if k % 10 == 9:  # scan-match style: broadcast z and quat_meas, broadcast diagonal R
            z4 = np.concatenate([0.1 * rng.standard_normal(3), [0.0]])
            qm = np.array([0.99, 0.0, 0.0, 0.14]); qm /= np.linalg.norm(qm)
            rep.update_indexed([9, 10, 11, 8], tile(z4), [0.0025] * 3 + [3e-4], quat_meas=tile(qm))
            bc.update_indexed([9, 10, 11, 8], z4, [0.0025] * 3 + [3e-4], quat_meas=qm)
    for a, b in zip(rep.get_head(), bc.get_head()):
        assert np.array_equal(a, b)
    with pytest.raises(ValueError):  # the binding refuses to mix a per-filter mask with broadcast blocks ...
        bc.step_legodo(imu, lo, np.ones(B, dtype=np.uint8), q4)
    import ctypes as C
    from pronto_amd._lib import PB_HOST_BROADCAST
    m8 = np.ones(B, dtype=np.uint8)  # ... and so does the C ABI
    rc = bc._L.pb_step_legodo(bc._h, C.c_void_p(imu.ctypes.data), C.c_void_p(lo.ctypes.data), C.c_void_p(m8.ctypes.data),
                              (C.c_double * 4)(*q4), PB_HOST_BROADCAST)
    assert rc != 0 and b"BROADCAST" in bc._L.pb_last_error(bc._h)
    rep.close(); bc.close()


@pytest.mark.parametrize("n", [15, 21])
def test_output_slot_checkpoints_equal_copies(pa, oracle, n):
    """pb_set_output_slot: an update that writes its posterior straight into a checkpoint slot must leave exactly what
    update + pb_state_save leaves, for every update kind; and a saved posterior is never modified by later updates."""
    B, T = 300, 12
    w = Workload(B, n_states=n)
    vec, quat, P0 = w.initial_state()
    q4 = w.process_noise()
    a = pa.BatchEstimator(B, n_states=n)   # update in place, then copy
    b = pa.BatchEstimator(B, n_states=n)   # update straight into the slot
    for e in (a, b):
        e.set_constants(*oracle.constants())
        e.reset(vec, quat, P0)
        e.history_reserve(3 * T)
    for k in range(T):
        imu = w.imu_block(k)
        lo, mask = w.legodo_block(k)
        z, qm, Rd = (w.scanmatch_block(k) if n == 21 else w.vo_block(k))
        idx = [9, 10, 11, 8] if n == 21 else [9, 10, 11, 6, 7, 8]
        a.predict(imu, q4); a.state_save(3 * k)
        b.set_output_slot(3 * k); b.predict(imu, q4); b.state_save(3 * k)          # save = no-op here
        a.step_legodo(imu, lo, mask, q4); a.state_save(3 * k + 1)
        b.set_output_slot(3 * k + 1); b.step_legodo(imu, lo, mask, q4)
        a.update_indexed(idx, pad_z(z, len(idx)), Rd, quat_meas=np.ascontiguousarray(qm))
        a.state_save(3 * k + 2)
        if k % 2:   # an update that finds the head in a slot and has no slot of its own goes back to the own array
            b.update_indexed(idx, pad_z(z, len(idx)), Rd, quat_meas=np.ascontiguousarray(qm))
            b.state_save(3 * k + 2)
        else:
            b.set_output_slot(3 * k + 2)
            b.update_indexed(idx, pad_z(z, len(idx)), Rd, quat_meas=np.ascontiguousarray(qm))
    for x, y in zip(a.get_head(), b.get_head()):
        assert np.array_equal(x, y)
    for slot in range(3 * T):              # every checkpoint, including the ones later updates read from
        a.state_restore(slot); b.state_restore(slot)
        for x, y in zip(a.get_head(), b.get_head()):
            assert np.array_equal(x, y), slot
    a.close(); b.close()


def test_config4_256k_whole_equals_its_eight_shards(pa, oracle):
    """BASELINE config 4 (262 144 filters sharded 8 x 32 768) rehearsed on ONE GPU: the whole batch in one context, then
    each of the eight filter-range shards alone (what rank r of `bench.py --gpus 8` runs: Workload(b0=...) + its own
    context).  Filters never interact, so (a) every shard's head state is BIT-identical to its slice of the whole batch,
    (b) the reduction of the eight shard summaries (what the end-of-run all-reduce computes) equals the whole batch's
    pb_summary, (c) sampled filters of the whole batch match the oracle, (d) nothing non-finite."""
    import torch
    from pronto_amd.shard import reduce_summaries, shard_range
    Btot, world, n, T = 262144, 8, 15, 16
    dev = torch.device("cuda:0")
    w = Workload(Btot, n_states=n)
    vec, quat, P0 = w.initial_state()
    q4 = w.process_noise()
    imu, lo, mask = w.streams(0, T)
    est = pa.BatchEstimator(Btot, n_states=n)
    est.set_constants(*oracle.constants())
    est.reset(vec, quat, P0)
    est.run_legodo(torch.from_numpy(imu).to(dev), torch.from_numpy(lo).to(dev), torch.from_numpy(mask).to(dev), q4)
    whole = est.summary()
    v, q, P, ll = est.get_head()
    est.close()
    assert whole[3] == 0 and whole[2] < 1e-12                                             # (d)
    sel = np.concatenate([np.arange(0, 32), np.arange(32768 - 16, 32768 + 16), np.arange(Btot - 32, Btot)])
    v21, P21 = embed21(vec[:, sel], P0[:, :, sel])
    ob = oracle.OracleBatch(v21, quat[:, sel], P21)
    ob.run_legodo(np.ascontiguousarray(imu[:, :, sel]), np.ascontiguousarray(lo[:, :, sel]),
                  np.ascontiguousarray(mask[:, sel]), q4)
    assert rel(v[:, sel], ob.vec[:n]) < TOL and rel(q[:, sel], ob.quat) < TOL             # (c)
    assert rel(P[:, :, sel], ob.cov[:n, :n]) < TOL and rel(ll[sel], ob.ll) < TOL
    shard_sums = []
    for r in range(world):
        b0, b1 = shard_range(Btot, r, world)
        assert b1 - b0 == 32768
        if r in (0, 5):   # exactly what a rank does: its own generator instance at offset b0
            ws = Workload(b1 - b0, b0=b0, n_states=n)
            sv, sq, sP = ws.initial_state()
            simu, slo, smask = ws.streams(0, T)
        else:             # the generator is counter-based (tests/test_shard_dist.py): a slice is the same bits
            sv, sq, sP = (np.ascontiguousarray(a[..., b0:b1]) for a in (vec, quat, P0))
            simu, slo, smask = (np.ascontiguousarray(a[..., b0:b1]) for a in (imu, lo, mask))
        e = pa.BatchEstimator(b1 - b0, n_states=n)
        e.set_constants(*oracle.constants())
        e.reset(sv, sq, sP)
        e.run_legodo(torch.from_numpy(simu).to(dev), torch.from_numpy(slo).to(dev), torch.from_numpy(smask).to(dev), q4)
        shard_sums.append(e.summary())
        hv, hq, hP, hl = e.get_head()
        e.close()
        assert np.array_equal(hv, v[:, b0:b1]) and np.array_equal(hq, q[:, b0:b1])       # (a)
        assert np.array_equal(hP, P[:, :, b0:b1]) and np.array_equal(hl, ll[b0:b1])
    red = reduce_summaries(shard_sums)                                                    # (b)
    assert np.isclose(red[0], whole[0], rtol=1e-12) and np.isclose(red[1], whole[1], rtol=1e-12)
    assert red[2] == whole[2] and red[3] == whole[3] == 0


@pytest.mark.parametrize("n,kind", [(15, 0), (15, 1), (21, 0), (21, 1)])
def test_fused_correction_step_equals_three_updates(pa, oracle, n, kind):
    """pb_step_legodo_correct (predict + leg-odometry m=3 + VO position_orient m=6 / scan-match position_yaw m=4 in ONE
    kernel and one state round trip) against the oracle's three updates and against the three-launch path
    (pb_step_legodo + pb_update_indexed_orient; equal to rounding, <= 1e-12: the generic update kernel scales W by
    |D|^-1/2, the fused one by D^-1).  Ragged batch, masks on both measurements, host / device / mixed memory spaces,
    per-filter and broadcast R."""
    import torch
    from pronto_amd._lib import PB_CORR_POS_ORIENT, PB_CORR_POS_YAW
    B, T = 333, 48
    dev = torch.device("cuda:0")
    w = Workload(B, n_states=n)
    est_f, ob = make_pair(pa, oracle, w, dense_p0=4)
    est_s, _ = make_pair(pa, oracle, w, dense_p0=4)
    q4 = w.process_noise()
    idx = [9, 10, 11, 6, 7, 8] if kind == 0 else [9, 10, 11, 8]
    ck = PB_CORR_POS_ORIENT if kind == 0 else PB_CORR_POS_YAW
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    for k in range(T):
        imu = w.imu_block(k)
        lo, mask = w.legodo_block(k)
        ob.predict(imu, q4)
        ob.update_indexed([3, 4, 5], lo[0:3], lo[3:6], mask=mask)
        if k % 3 != 2:
            est_f.step_legodo(imu, lo, mask, q4)
            est_s.step_legodo(imu, lo, mask, q4)
            continue
        z, qm, Rd = w.vo_block(k) if kind == 0 else w.scanmatch_block(k)
        zz, qm, Rd = pad_z(z, len(idx)), np.ascontiguousarray(qm), np.ascontiguousarray(Rd)
        mask2 = ((np.arange(B) + k) % 7 != 0).astype(np.uint8)
        ob.update_indexed(idx, zz, Rd, quat_meas=qm, mask=mask2)
        est_s.step_legodo(imu, lo, mask, q4)
        est_s.update_indexed(idx, zz, Rd, mask=mask2, quat_meas=qm)
        v = (k // 3) % 3
        if v == 0:      # everything from the host
            est_f.step_legodo_correct(imu, lo, mask, q4, ck, zz, Rd, qm, mask2)
        elif v == 1:    # IMU / leg odometry from the host, the correction resident on the device (FovisHandler's case)
            est_f.step_legodo_correct(imu, lo, mask, q4, ck, up(zz), up(Rd), up(qm), up(mask2))
        else:           # everything on the device
            est_f.step_legodo_correct(up(imu), up(lo), up(mask), q4, ck, up(zz), up(Rd), up(qm), up(mask2))
    check(est_f, ob)
    check(est_s, ob)
    for a, b in zip(est_f.get_head(), est_s.get_head()):
        assert rel(a, b) < 1e-12
    # broadcast R2 (a handler's cov_* members) == the same values per filter, bit for bit
    e1, _ = make_pair(pa, oracle, w)
    e2, _ = make_pair(pa, oracle, w)
    imu = w.imu_block(0); lo, mask = w.legodo_block(0)
    z, qm, Rd = w.vo_block(0) if kind == 0 else w.scanmatch_block(0)
    zz, qm = pad_z(z, len(idx)), np.ascontiguousarray(qm)
    rb = [float(Rd[i, 0]) for i in range(len(idx))]
    e1.step_legodo_correct(imu, lo, mask, q4, ck, zz, np.ascontiguousarray(np.tile(np.array(rb)[:, None], (1, B))), qm)
    e2.step_legodo_correct(imu, lo, mask, q4, ck, zz, rb, qm)
    for a, b in zip(e1.get_head(), e2.get_head()):
        assert np.array_equal(a, b)
    with pytest.raises(ValueError):
        e1.step_legodo_correct(imu, lo, mask, q4, 7, zz, rb, qm)     # unknown correction kind: refused by the binding ...
    import ctypes as C
    rc = e1._L.pb_step_legodo_correct(e1._h, C.c_void_p(imu.ctypes.data), C.c_void_p(lo.ctypes.data), None,
                                      (C.c_double * 4)(*q4), 0, 7, C.c_void_p(zz.ctypes.data),
                                      C.c_void_p(np.array(rb).ctypes.data), 0, C.c_void_p(qm.ctypes.data), None, 0)
    assert rc == 1 and b"corr_kind" in e1._L.pb_last_error(e1._h)  # ... and by the C ABI


@pytest.mark.parametrize("n", [15, 21])
def test_handler_index_lists_on_the_cooperative_update_kernel(pa, oracle, n, monkeypatch):
    """Every index list a handler produces (velocity, position, position+velocity, position+orientation, position+yaw,
    velocity+yaw, yaw) as a stand-alone update: the compile-time-index cooperative kernel against the oracle and against
    the generic run-time-index kernel (<= 1e-12), per-filter and broadcast diagonal R, masks; a full R or any other index
    list still takes the generic kernel."""
    from oracle import numpy_restatement as nr
    B = 200
    rng = np.random.default_rng(77 + n)
    w = Workload(B, n_states=n)
    lists = [([3, 4, 5], False), ([9, 10, 11], False), ([9, 10, 11, 3, 4, 5], False), ([9, 10, 11, 6, 7, 8], True),
             ([9, 10, 11, 8], True), ([3, 4, 5, 8], True), ([8], True), ([9, 10, 11], True), ([4, 9], False),
             ([3, 4, 5, 0, 1, 2], False)]   # LegOdoCommon lin_rot_rate: one-lane in-register kernel for 15 states
    ests = []
    for gen in ("0", "1"):
        monkeypatch.setenv("PRONTO_BATCH_GENERIC_UPDATE", gen)
        est, ob = make_pair(pa, oracle, w, dense_p0=9)
        ests.append(est)
    q4 = w.process_noise()
    for t, (idx, orient) in enumerate(lists * 2):
        imu = w.imu_block(t)
        for e in ests:
            e.predict(imu, q4)
        ob.predict(imu, q4)
        m = len(idx)
        z = np.ascontiguousarray(ob.vec[idx] + 0.05 * rng.normal(size=(m, B)))
        mask = (rng.random(B) > 0.25).astype(np.uint8)
        qm = None
        if orient:
            d = 0.03 * rng.normal(size=(3, B))
            qm = np.ascontiguousarray(nr.quat_mul(ob.quat.T, nr.quat_exp(d.T)[0]).T)
        if t < len(lists):
            Rd = np.ascontiguousarray(0.01 + 0.05 * rng.random((m, B)))
            for e in ests:
                e.update_indexed(idx, z, Rd, mask=mask, quat_meas=qm)
        else:
            rb = list(0.01 + 0.05 * rng.random(m))
            Rd = np.tile(np.array(rb)[:, None], (1, B))
            for e in ests:
                e.update_indexed(idx, z, rb, mask=mask, quat_meas=qm)
        ob.update_indexed(idx, z, Rd, quat_meas=qm, mask=mask)
    for e in ests:
        check(e, ob)
    for a, b in zip(ests[0].get_head(), ests[1].get_head()):
        assert rel(a, b) < 1e-12


@pytest.mark.parametrize("n", [15, 21])
def test_split_space_step_equals_the_plain_step(pa, oracle, n):
    """pb_step_legodo_split: the IMU block and the leg-odometry block (+ mask) in different memory spaces -- host / device,
    device / host, broadcast / device, device / broadcast -- must give bit-identical results to the same numbers passed
    as per-filter device blocks."""
    import torch
    B = 300
    dev = torch.device("cuda:0")
    w = Workload(B, n_states=n)
    vec, quat, P0 = w.initial_state()
    q4 = w.process_noise()
    ref = pa.BatchEstimator(B, n_states=n)
    alt = pa.BatchEstimator(B, n_states=n)
    for e in (ref, alt):
        e.set_constants(*oracle.constants())
        e.reset(vec, quat, P0)
    tile = lambda a: np.ascontiguousarray(np.repeat(np.asarray(a, dtype=np.float64)[:, None], B, axis=1))
    for k in range(24):
        imu = w.imu_block(k)
        lo, mask = w.legodo_block(k)
        case = k % 4
        if case == 2:   # one IMU message for everybody
            imu = tile(imu[:, 0])
        if case == 3:   # one measurement for everybody, no mask
            lo, mask = tile(lo[:, 0]), None
        d_imu, d_lo = torch.from_numpy(imu).to(dev), torch.from_numpy(lo).to(dev)
        d_mask = None if mask is None else torch.from_numpy(mask).to(dev)
        ref.step_legodo(d_imu, d_lo, d_mask, q4)
        if case == 0:
            alt.step_legodo(imu, d_lo, d_mask, q4)                                   # host / device
        elif case == 1:
            alt.step_legodo(d_imu, lo, mask, q4)                                     # device / host
        elif case == 2:
            alt.step_legodo(np.ascontiguousarray(imu[:, 0]), d_lo, d_mask, q4)       # broadcast / device
        else:
            alt.step_legodo(d_imu, np.ascontiguousarray(lo[:, 0]), None, q4)         # device / broadcast
    for a, b in zip(ref.get_head(), alt.get_head()):
        assert np.array_equal(a, b)
    ref.close(); alt.close()


@pytest.mark.parametrize("n", [15, 21])
def test_lin_rot_rate_compile_time_and_run_time_lists_agree(pa, oracle, monkeypatch, n):
    """LegOdoCommon's lin_rot_rate list [3,4,5,0,1,2] (rbis_legodo_common.cpp:66-67).  15 states: the compile-time-list
    kernel (k_update_lane, one lane per filter) and the run-time-list kernel (PRONTO_BATCH_GENERIC_UPDATE=1: six indices run
    on two waves per tile, k_update_coop_rt) agree to rounding (the two-wave kernel scales W by 1/d once, like the
    four-wave ones).  21 states: k_update_quad_list (the four-wave body with the list folded in at compile time) and
    k_update_quad_rt are the same arithmetic: bit-identical.  Both against the oracle."""
    B, idx = 300, [3, 4, 5, 0, 1, 2]
    rng = np.random.default_rng(21)
    w = Workload(B, n_states=n)
    ests = []
    for gen in ("0", "1"):
        monkeypatch.setenv("PRONTO_BATCH_GENERIC_UPDATE", gen)
        est, ob = make_pair(pa, oracle, w, dense_p0=4)
        ests.append(est)
    q4 = w.process_noise()
    for t in range(6):
        imu = w.imu_block(t)
        for e in ests:
            e.predict(imu, q4)
        ob.predict(imu, q4)
        z = np.ascontiguousarray(ob.vec[idx] + 0.05 * rng.normal(size=(6, B)))
        mask = (rng.random(B) > 0.2).astype(np.uint8)
        if t % 2:
            rb = list(0.01 + 0.05 * rng.random(6))
            Rd = np.tile(np.array(rb)[:, None], (1, B))
            for e in ests:
                e.update_indexed(idx, z, rb, mask=mask)
        else:
            Rd = np.ascontiguousarray(0.01 + 0.05 * rng.random((6, B)))
            for e in ests:
                e.update_indexed(idx, z, Rd, mask=mask)
        ob.update_indexed(idx, z, Rd, mask=mask)
    for e in ests:
        check(e, ob)
    for a, b in zip(ests[0].get_head(), ests[1].get_head()):
        assert rel(a, b) < 1e-12
        assert n == 15 or np.array_equal(a, b)

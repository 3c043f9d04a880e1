"""Parity hardening on the GPU (-m gpu): the recalled eigen_utils constants as switches, and the reference's edge branches
that a well-behaved synthetic workload never reaches -- chi kept in the vector below chiToQuat's tolerance, orientation
residuals near pi and negated measurement quaternions (subtractQuats' wrap), dt = 0 and dt > 0.1 s steps, measurement
covariances twelve orders of magnitude apart.  Every case: HIP path through the C ABI against the oracle, block-relative
<= 1e-9 AND element-wise (tests/util.py rel_elem)."""
import numpy as np
import pytest

from test_gpu_parity import check, make_pair, pa  # noqa: F401  (pa is a fixture)
from util import pad_z, rel

from pronto_amd.synth import Workload, _quat_exp, _quat_mul

pytestmark = pytest.mark.gpu


@pytest.fixture()
def constants_restored(oracle):
    g, tol = oracle.constants()
    yield oracle
    oracle.lib().po_set_constants(g, tol)


@pytest.mark.parametrize("n", [15, 21])
@pytest.mark.parametrize("g,tol", [(9.8, 1e-7), (9.81, 1e-5), (9.80665, 0.0)])
def test_alternative_recalled_constants(pa, constants_restored, n, g, tol):
    """g_val and chiToQuat's tolerance are not in the reference tree (recalled).  Both sides take them as
    run-time constants: SURVEY's recollection (9.8, 1e-7), another plausible pair, and tolerance 0 (always fold) must all
    agree between the HIP path and the oracle -- so whichever value eigen_utils really has, it is a one-line change."""
    oracle = constants_restored
    oracle.lib().po_set_constants(g, tol)
    assert oracle.constants() == (g, tol)
    B, T = 130, 90
    w = Workload(B, n_states=n)
    est, ob = make_pair(pa, oracle, w, dense_p0=2)      # make_pair hands oracle.constants() to pb_set_constants
    q4 = w.process_noise()
    for k in range(T):
        imu = w.imu_block(k)
        lo, mask = w.legodo_block(k)
        est.step_legodo(imu, lo, mask, q4)
        ob.predict(imu, q4)
        ob.update_indexed([3, 4, 5], lo[0:3], lo[3:6], mask=mask)
        if k % 16 == 15:
            z, qm, Rd = w.vo_block(k)
            est.update_indexed([9, 10, 11, 6, 7, 8], pad_z(z, 6), Rd, quat_meas=np.ascontiguousarray(qm))
            ob.update_indexed([9, 10, 11, 6, 7, 8], pad_z(z, 6), Rd, quat_meas=qm)
    check(est, ob)
    # and the constants do matter: the default pair gives a different trajectory
    est2 = pa.BatchEstimator(B, n_states=n)
    est2.set_constants(9.80665 if g != 9.80665 else 9.8, 1e-6)
    vec, quat, P0 = w.initial_state()
    est2.reset(vec, quat, P0)
    est2.step_legodo(w.imu_block(0), *w.legodo_block(0), q4)
    est.reset(vec, quat, P0)
    est.step_legodo(w.imu_block(0), *w.legodo_block(0), q4)
    if g != 9.80665:
        assert rel(est2.get_head()[0], est.get_head()[0]) > 1e-9


@pytest.mark.parametrize("n", [15, 21])
def test_chi_below_fold_tolerance_is_carried_in_the_vector(pa, oracle, n):
    """chiToQuat folds chi into the quaternion only when |chi| > tol (eigen_utils, recalled: rbis.cpp:63-69 call site);
    below it chi STAYS in vec[6:9] and accumulates.  Rates chosen so that |omega| dt straddles the tolerance per filter
    (0.05 ... 20 x tol): 200 predict steps with fused velocity updates every 4th; chi entries compared one by one."""
    g, tol = oracle.constants()
    B, T, dt = 128, 200, 1e-3
    w = Workload(B, n_states=n)
    est, ob = make_pair(pa, oracle, w)
    q4 = w.process_noise()
    rng = np.random.default_rng(3)
    scale = tol / dt * np.logspace(np.log10(0.05), np.log10(20.0), B)        # |omega| per filter
    axis = rng.normal(size=(3, B))
    axis /= np.linalg.norm(axis, axis=0)
    below = above = 0
    for k in range(T):
        imu = w.imu_block(k)
        imu[0:3] = axis * scale * (1.0 + 0.3 * np.sin(0.1 * k))
        if n == 21:
            imu[0:3] += ob.vec[15:18]       # the filter subtracts its bias estimate: keep omega where we want it
        imu[6] = dt
        lo, mask = w.legodo_block(k)
        if k % 4 == 3:
            est.step_legodo(imu, lo, mask, q4)
            ob.predict(imu, q4)
            ob.update_indexed([3, 4, 5], lo[0:3], lo[3:6], mask=mask)
        else:
            est.predict(imu, q4)
            ob.predict(imu, q4)
        chi = np.linalg.norm(ob.vec[6:9], axis=0)
        below += int(np.sum((chi > 0) & (chi <= tol)))
        above += int(np.sum(chi == 0))
    assert below > 1000 and above > 1000            # both branches were taken many times
    check(est, ob)
    v = est.get_head()[0]
    assert np.max(np.abs(v[6:9] - ob.vec[6:9])) < 1e-18 + 1e-9 * tol


@pytest.mark.parametrize("n,idx", [(15, [9, 10, 11, 6, 7, 8]), (21, [9, 10, 11, 8]), (15, [8]), (21, [9, 10, 11, 6, 7, 8])])
@pytest.mark.parametrize("generic", ["0", "1"])
def test_orientation_residual_near_pi_and_negated_quaternion(pa, oracle, n, idx, generic, monkeypatch):
    """subtractQuats (eigen_utils, call site rbis.cpp:199) returns the rotation vector of q^-1 q_meas with the angle wrapped
    into [-pi, pi]: measurement quaternions with the OPPOSITE sign (w < 0) and residual angles 0.9 pi ... pi - 1e-9, on the
    cooperative and on the generic update kernel, and (n = 15) inside the fused correction step.  (At exactly pi the
    relative quaternion's w is 0 to rounding and its SIGN -- hence the sign of the whole residual -- is decided by the last
    bit of a quaternion product: no two implementations, the reference's included, agree there; pi - 1e-9 is the closest
    well-posed case.)"""
    monkeypatch.setenv("PRONTO_BATCH_GENERIC_UPDATE", generic)
    B = 96
    w = Workload(B, n_states=n)
    est, ob = make_pair(pa, oracle, w)
    rng = np.random.default_rng(12)
    angles = np.concatenate([np.linspace(0.9 * np.pi, np.pi - 1e-6, 40), [np.pi - 1e-7, np.pi - 1e-8, np.pi - 1e-9],
                             np.linspace(0.0, 0.2, B - 43)])
    axis = rng.normal(size=(3, B))
    axis /= np.linalg.norm(axis, axis=0)
    q4 = w.process_noise()
    m = len(idx)
    for trial in range(3):
        imu = w.imu_block(trial)
        est.predict(imu, q4)
        ob.predict(imu, q4)
        qm = _quat_mul(ob.quat, _quat_exp(axis * angles))
        if trial >= 1:
            qm = -qm                                   # the same rotation, opposite sign
        if trial == 2:
            qm[:, ::2] *= -1.0                         # mixed signs across the batch
        qm = np.ascontiguousarray(qm)
        z = np.ascontiguousarray(np.vstack([ob.vec[[i for i in idx if i < 6 or i > 8]] + 0.02 * rng.normal(size=(m - sum(6 <= i <= 8 for i in idx), B)),
                                            np.zeros((sum(6 <= i <= 8 for i in idx), B))]))
        Rd = np.ascontiguousarray(np.full((m, B), 0.05 ** 2))
        est.update_indexed(idx, z, Rd, quat_meas=qm)
        ob.update_indexed(idx, z, Rd, quat_meas=qm)
        check(est, ob)
    if n == 15 and m in (4, 6) and generic == "0":
        from pronto_amd._lib import PB_CORR_POS_ORIENT, PB_CORR_POS_YAW
        imu = w.imu_block(5)
        lo, mask = w.legodo_block(5)
        ob.predict(imu, q4)
        ob.update_indexed([3, 4, 5], lo[0:3], lo[3:6], mask=mask)
        qm = np.ascontiguousarray(-_quat_mul(ob.quat, _quat_exp(axis * angles)))
        est.step_legodo_correct(imu, lo, mask, q4, PB_CORR_POS_ORIENT if m == 6 else PB_CORR_POS_YAW, z, Rd, qm)
        ob.update_indexed(idx, z, Rd, quat_meas=qm)
        check(est, ob)


@pytest.mark.parametrize("n", [15, 21])
def test_dt_zero_and_dt_above_a_tenth_of_a_second(pa, oracle, n):
    """integration_dt comes from message time stamps (sensor_handlers.cpp:239-249): a repeated stamp gives dt = 0 (the
    covariance still gets its omega / accel diagonal blocks overwritten, rbis.cpp:120-121), a gap gives dt = 0.2 s (the
    reference only warns above 0.1 s, :245-248).  Per-filter dt in {0, 1e-3, 0.2} within one batch."""
    B = 192
    w = Workload(B, n_states=n)
    est, ob = make_pair(pa, oracle, w, dense_p0=3)
    q4 = w.process_noise()
    dts = np.tile(np.array([0.0, 1e-3, 0.2]), B // 3)
    for k in range(24):
        imu = w.imu_block(k)
        imu[6] = np.roll(dts, k)
        lo, mask = w.legodo_block(k)
        if k % 2:
            est.step_legodo(imu, lo, mask, q4)
            ob.predict(imu, q4)
            ob.update_indexed([3, 4, 5], lo[0:3], lo[3:6], mask=mask)
        else:
            est.predict(imu, q4)
            ob.predict(imu, q4)
    check(est, ob)


@pytest.mark.parametrize("n", [15, 21])
@pytest.mark.parametrize("full_r", [False, True])
def test_measurement_covariances_from_1e_minus_12_to_1e_plus_6(pa, oracle, n, full_r):
    """R entries of 1e-12 (the measurement pins the state) and 1e+6 (it is ignored), mixed per filter and per row, in the
    fused step (m = 3) and in the correction (m = 6 / 4) on the cooperative kernel or -- the same diagonal passed as a full
    column-major R -- on the generic kernel; then ordinary steps on the pinned covariance.  ONE extreme update per state
    block: P - K C P (rbis.cpp:140,226, not Joseph) loses eleven digits each time R is 1e-12, so a second pin of the same
    state is decided by rounding in any implementation, the reference's included."""
    B = 128
    rng = np.random.default_rng(21 + n)
    w = Workload(B, n_states=n)
    est, ob = make_pair(pa, oracle, w)
    q4 = w.process_noise()
    pick = lambda shape: np.ascontiguousarray(np.choose(rng.integers(0, 4, size=shape), [1e-12, 1e-4, 1.0, 1e6]))
    idx = [9, 10, 11, 6, 7, 8] if n == 15 else [9, 10, 11, 8]
    m = len(idx)
    for k in range(10):
        imu = w.imu_block(k)
        lo, mask = w.legodo_block(k)
        if k == 0:
            lo[3:6] = pick((3, B))
        est.step_legodo(imu, lo, mask, q4)
        ob.predict(imu, q4)
        ob.update_indexed([3, 4, 5], lo[0:3], lo[3:6], mask=mask)
        if k in (0, 5):
            z, qm, Rd = w.vo_block(k) if n == 15 else w.scanmatch_block(k)
            if k == 0:
                Rd = pick((m, B))
            if full_r:
                Rf = np.zeros((m * m, B))
                for i in range(m):
                    Rf[i * m + i] = Rd[i]
                est.update_indexed(idx, pad_z(z, m), np.ascontiguousarray(Rf), quat_meas=np.ascontiguousarray(qm))
            else:
                est.update_indexed(idx, pad_z(z, m), np.ascontiguousarray(Rd), quat_meas=np.ascontiguousarray(qm))
            ob.update_indexed(idx, pad_z(z, m), Rd, quat_meas=qm)
        if k == 0:
            check(est, ob)                 # the extreme updates themselves: full tolerance
    check(est, ob, tol=1e-7)               # ten steps on: the pinned blocks have lost digits on BOTH sides


@pytest.mark.parametrize("n", [15, 21])
def test_store_hazard_regression_thousands_of_launches(pa, n):
    """a buffer_store_dwordx4 with an SGPR soffset directly followed by a VALU write of one of its data registers
    stored the NEW value in 16 lanes of the row about once per 1000 launches, until every 16-byte store got an `s_nop 1` tied
    to its data registers (rbis_kernels.hpp stg2).  The guard: 5 000 launches of the hot step kernel (k_step_coop<15> /
    k_step_quad for 21 states) plus 1 000 of the tile copy (k_calib_copy) on 64k filters; the same 50-step replay must give
    the same bits -- every 64-bit word of the 73 / 135 MB state -- 100 times over (pb_state_checksum).
    (scripts/chk_store_hazard.py checks the same property statically on the ISA; tests/test_isa_hazard.py runs it.)"""
    import torch
    B, K, REPS = 65536, 50, 100
    dev = torch.device("cuda:0")
    w = Workload(B, n_states=n)
    vec, quat, P0 = w.initial_state()
    q4 = w.process_noise()
    est = pa.BatchEstimator(B, n_states=n)
    est.history_reserve(1)
    est.reset(vec, quat, P0)
    est.state_save(0)
    blocks = [(torch.from_numpy(w.imu_block(k)).to(dev), *[torch.from_numpy(a).to(dev) for a in w.legodo_block(k)]) for k in range(K)]
    kernel = est.hot_kernel() if hasattr(est, "hot_kernel") else ""
    first = None
    for rep in range(REPS):
        est.state_restore(0)
        for imu, lo, mask in blocks:
            est.step_legodo(imu, lo, mask, q4)
        cs = est.state_checksum()
        assert est.calib_copy_checksum(10) == cs, rep   # k_calib_copy: a tile copy with the step kernels' 16-byte accesses
        if first is None:
            first = cs
        assert cs == first, (rep, kernel)
    s = est.summary()
    assert s[3] == 0 and np.isfinite(s[0])
    est.close()

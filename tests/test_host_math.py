"""The kernels' per-lane arithmetic (pronto_amd/csrc/rbis_device.hpp), compiled for the host by the test-only
harness, against the oracle.  Catches maths errors in a container without a GPU; the real parity tests are the
-m gpu ones, which run the same header through hipcc and the C ABI."""
import ctypes as C

import numpy as np
import pytest

from util import embed21, pad_z, random_spd, rel

from pronto_amd.synth import Workload

DP = C.POINTER(C.c_double)


def P(a):
    return a.ctypes.data_as(DP)


def pack(H, ns, vec, quat, cov, ll):
    B = vec.shape[1]
    st = np.zeros((H.hh_nc(ns), B))
    st[:ns] = vec[:ns]
    st[ns:ns + 4] = quat
    st[ns + 4] = ll
    for i in range(ns):
        for j in range(i + 1):
            st[ns + 5 + H.hh_pk(i, j)] = cov[i, j]
    return st


def unpack(H, ns, st):
    B = st.shape[1]
    cov = np.zeros((ns, ns, B))
    for i in range(ns):
        for j in range(i + 1):
            cov[i, j] = cov[j, i] = st[ns + 5 + H.hh_pk(i, j)]
    return st[:ns], st[ns:ns + 4], cov, st[ns + 4]


def quat_mul(a, b):
    """Hamilton product of [4, B] quaternion arrays (w first)."""
    w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3]
    x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2]
    y = a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3]
    z = a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1]
    return np.stack([w, x, y, z])


@pytest.mark.parametrize("ns", [15, 21])
def test_structured_math_matches_dense_oracle(oracle, harness, ns):
    H = harness
    g, tol = oracle.constants()
    B, T = 32, 400
    w = Workload(B, n_states=ns)
    vec, quat, P0 = w.initial_state()
    P0 = P0 + random_spd(ns, B, 0.03, 11)   # dense P0: exercises the omega/accel cross blocks too
    if ns == 21:
        vec[15:18], vec[18:21] = 0.5 * w.bg, 0.5 * w.ba
    v21, P21 = embed21(vec, P0)
    ob = oracle.OracleBatch(v21, quat, P21)
    st = pack(H, ns, vec, quat, P0, np.zeros(B))
    q4 = w.process_noise()
    gd, td = C.c_double(g), C.c_double(tol)
    for k in range(T):
        imu = w.imu_block(k)
        lo, mask = w.legodo_block(k)
        ob.predict(imu, q4)
        ob.update_indexed([3, 4, 5], lo[0:3], lo[3:6], mask=mask)
        H.hh_step(ns, P(st), C.c_long(B), B, P(imu), P(lo), mask.ctypes.data_as(C.c_void_p), P(q4), gd, td, 1)
        if k % 32 == 31:
            z, qm, Rd = w.vo_block(k)
            zz, qm = pad_z(z, 6), np.ascontiguousarray(qm)
            ob.update_indexed([9, 10, 11, 6, 7, 8], zz, Rd, quat_meas=qm)
            H.hh_update_vo(ns, P(st), C.c_long(B), B, P(zz), P(Rd), P(qm), gd, td)
        if k % 25 == 24:
            z, qm, Rd = w.scanmatch_block(k)
            zz, qm = pad_z(z, 4), np.ascontiguousarray(qm)
            ob.update_indexed([9, 10, 11, 8], zz, Rd, quat_meas=qm)
            H.hh_update_posyaw(ns, P(st), C.c_long(B), B, P(zz), P(Rd), P(qm), gd, td)
    v, q, cov, ll = unpack(H, ns, st)
    # tolerance: fp64 reassociation only (structured vs dense), accumulated over 400 steps
    assert rel(v, ob.vec[:ns]) < 1e-11
    assert rel(q, ob.quat) < 1e-11
    assert rel(cov, ob.cov[:ns, :ns]) < 1e-11
    assert rel(ll, ob.ll) < 1e-11


@pytest.mark.parametrize("ns", [15, 21])
@pytest.mark.parametrize("upd", [1, 0])
def test_cooperative_roles_match_oracle(oracle, harness, ns, upd):
    """rbis_coop.hpp: role C (core sub-matrix) + role P (passive panels) with the LDS hand-off replaced by an array."""
    H = harness
    g, tol = oracle.constants()
    B, T = 24, 150
    w = Workload(B, n_states=ns)
    vec, quat, P0 = w.initial_state()
    P0 = P0 + random_spd(ns, B, 0.03, 5)
    if ns == 21:
        vec[15:18], vec[18:21] = 0.5 * w.bg, 0.5 * w.ba
    v21, P21 = embed21(vec, P0)
    ob = oracle.OracleBatch(v21, quat, P21)
    st = pack(H, ns, vec, quat, P0, np.zeros(B))
    q4 = w.process_noise()
    for k in range(T):
        imu = w.imu_block(k)
        lo, mask = w.legodo_block(k)
        ob.predict(imu, q4)
        if upd:
            ob.update_indexed([3, 4, 5], lo[0:3], lo[3:6], mask=mask)
        H.hh_step_coop(ns, P(st), C.c_long(B), B, P(imu), P(lo), mask.ctypes.data_as(C.c_void_p), P(q4),
                       C.c_double(g), C.c_double(tol), upd)
    v, q, cov, ll = unpack(H, ns, st)
    assert rel(v, ob.vec[:ns]) < 1e-11 and rel(q, ob.quat) < 1e-11 and rel(cov, ob.cov[:ns, :ns]) < 1e-11
    if upd:
        assert rel(ll, ob.ll) < 1e-11


@pytest.mark.parametrize("upd", [1, 0])
def test_four_wave_roles_match_oracle(oracle, harness, upd):
    """rbis_quad.hpp: the 21-state step as four roles (P_cc | P_cb P_bb | omega column | accel column) run as four threads
    with a real barrier; the c-b coupling enters P_cc as the additive term H."""
    H = harness
    g, tol = oracle.constants()
    ns, B, T = 21, 24, 150
    w = Workload(B, n_states=ns)
    vec, quat, P0 = w.initial_state()
    P0 = P0 + random_spd(ns, B, 0.03, 5)
    vec[15:18], vec[18:21] = 0.5 * w.bg, 0.5 * w.ba
    ob = oracle.OracleBatch(vec, quat, P0)
    st = pack(H, ns, vec, quat, P0, np.zeros(B))
    q4 = w.process_noise()
    for k in range(T):
        imu = w.imu_block(k)
        lo, mask = w.legodo_block(k)
        ob.predict(imu, q4)
        if upd:
            ob.update_indexed([3, 4, 5], lo[0:3], lo[3:6], mask=mask)
        H.hh_step_quad(P(st), C.c_long(B), B, P(imu), P(lo), mask.ctypes.data_as(C.c_void_p), P(q4),
                       C.c_double(g), C.c_double(tol), upd)
    v, q, cov, ll = unpack(H, ns, st)
    assert rel(v, ob.vec[:ns]) < 1e-11 and rel(q, ob.quat) < 1e-11 and rel(cov, ob.cov[:ns, :ns]) < 1e-11
    from util import rel_elem
    assert rel_elem(cov, ob.cov[:ns, :ns]) < 1e-9
    if upd:
        assert rel(ll, ob.ll) < 1e-11


@pytest.mark.parametrize("alone", [False, True])
@pytest.mark.parametrize("ns,kind", [(15, 0), (15, 1), (21, 0), (21, 1)])
def test_cooperative_roles_with_fused_correction_match_oracle(oracle, harness, ns, kind, alone):
    """predict + leg-odometry + a second orientation update (VO position_orient m=6 / scan-match position_yaw m=4) through
    the two roles with both LDS hand-offs, against three separate oracle updates; second-update mask exercised."""
    H = harness
    g, tol = oracle.constants()
    B, T = 24, 60
    w = Workload(B, n_states=ns)
    vec, quat, P0 = w.initial_state()
    P0 = P0 + random_spd(ns, B, 0.03, 7)
    if ns == 21:
        vec[15:18], vec[18:21] = 0.5 * w.bg, 0.5 * w.ba
    v21, P21 = embed21(vec, P0)
    ob = oracle.OracleBatch(v21, quat, P21)
    st = pack(H, ns, vec, quat, P0, np.zeros(B))
    q4 = w.process_noise()
    idx = [9, 10, 11, 6, 7, 8] if kind == 0 else [9, 10, 11, 8]
    for k in range(T):
        imu = w.imu_block(k)
        lo, mask = w.legodo_block(k)
        z, qm, Rd = w.vo_block(k) if kind == 0 else w.scanmatch_block(k)
        zz, qm, Rd = pad_z(z, len(idx)), np.ascontiguousarray(qm), np.ascontiguousarray(Rd)
        mask2 = ((np.arange(B) + k) % 5 != 0).astype(np.uint8)
        ob.predict(imu, q4)
        ob.update_indexed([3, 4, 5], lo[0:3], lo[3:6], mask=mask)
        ob.update_indexed(idx, zz, Rd, quat_meas=qm, mask=mask2)
        if alone:   # the same three updates as two calls: fused step, then the correction ALONE on the two-role mapping
            H.hh_step_coop(ns, P(st), C.c_long(B), B, P(imu), P(lo), mask.ctypes.data_as(C.c_void_p), P(q4),
                           C.c_double(g), C.c_double(tol), 1)
        H.hh_step_coop_correct(ns, kind, 2 if alone else 1, P(st), C.c_long(B), B, P(imu), P(lo),
                               mask.ctypes.data_as(C.c_void_p), P(q4), C.c_double(g), C.c_double(tol), P(zz), P(Rd), P(qm),
                               mask2.ctypes.data_as(C.c_void_p))
    v, q, cov, ll = unpack(H, ns, st)
    assert rel(v, ob.vec[:ns]) < 1e-11 and rel(q, ob.quat) < 1e-11 and rel(cov, ob.cov[:ns, :ns]) < 1e-11
    assert rel(ll, ob.ll) < 1e-11


@pytest.mark.parametrize("kind", range(10))
def test_four_wave_stand_alone_updates_match_oracle(oracle, harness, kind):
    """rbis_quad.hpp quad_upd_*: the handlers' seven index lists as stand-alone updates on the four-wave mapping (one
    barrier), between four-wave steps, against the oracle's indexed (+ orientation) update; masks exercised."""
    H = harness
    g, tol = oracle.constants()
    ns, B, T = 21, 24, 40
    idx = [[3, 4, 5], [9, 10, 11], [9, 10, 11, 3, 4, 5], [9, 10, 11, 6, 7, 8], [9, 10, 11, 8], [3, 4, 5, 8], [8],
           [8, 9, 10, 11], [6, 7, 8, 9, 10, 11], [11]][kind]
    orient = 3 <= kind <= 6
    m = len(idx)
    w = Workload(B, n_states=ns)
    vec, quat, P0 = w.initial_state()
    P0 = P0 + random_spd(ns, B, 0.03, 7)
    vec[15:18], vec[18:21] = 0.5 * w.bg, 0.5 * w.ba
    ob = oracle.OracleBatch(vec, quat, P0)
    st = pack(H, ns, vec, quat, P0, np.zeros(B))
    q4 = w.process_noise()
    rng = np.random.default_rng(kind)
    for k in range(T):
        imu = w.imu_block(k)
        lo, mask = w.legodo_block(k)
        ob.predict(imu, q4)
        ob.update_indexed([3, 4, 5], lo[0:3], lo[3:6], mask=mask)
        H.hh_step_quad(P(st), C.c_long(B), B, P(imu), P(lo), mask.ctypes.data_as(C.c_void_p), P(q4),
                       C.c_double(g), C.c_double(tol), 1)
        # a measurement near the current estimate: z = x[idx] + noise, quaternion = head * small rotation
        zz = np.ascontiguousarray(ob.vec[idx, :] + 0.02 * rng.standard_normal((m, B)))
        dq = np.concatenate([np.ones((1, B)), 0.01 * rng.standard_normal((3, B))])
        qm = np.ascontiguousarray(quat_mul(ob.quat, dq / np.linalg.norm(dq, axis=0)))
        Rd = np.ascontiguousarray(np.abs(0.01 + 0.01 * rng.standard_normal((m, B))))
        mask2 = ((np.arange(B) + k) % 5 != 0).astype(np.uint8)
        ob.update_indexed(idx, zz, Rd, quat_meas=qm if orient else None, mask=mask2)
        H.hh_update_quad(kind, P(st), C.c_long(B), B, P(zz), P(Rd), P(qm), mask2.ctypes.data_as(C.c_void_p),
                         C.c_double(g), C.c_double(tol))
    v, q, cov, ll = unpack(H, ns, st)
    assert rel(v, ob.vec[:ns]) < 1e-11 and rel(q, ob.quat) < 1e-11 and rel(cov, ob.cov[:ns, :ns]) < 1e-11
    assert rel(ll, ob.ll) < 1e-11


def test_predict_only_and_masked_lanes(oracle, harness):
    H = harness
    g, tol = oracle.constants()
    B, ns = 16, 15
    w = Workload(B, n_states=ns)
    vec, quat, P0 = w.initial_state()
    v21, P21 = embed21(vec, P0)
    ob = oracle.OracleBatch(v21, quat, P21)
    st = pack(H, ns, vec, quat, P0, np.zeros(B))
    q4 = w.process_noise()
    mask = (np.arange(B) % 2).astype(np.uint8)   # every other filter's handler "returned NULL"
    for k in range(20):
        imu = w.imu_block(k)
        lo, _ = w.legodo_block(k)
        ob.predict(imu, q4)
        ob.update_indexed([3, 4, 5], lo[0:3], lo[3:6], mask=mask)
        H.hh_step(ns, P(st), C.c_long(B), B, P(imu), P(lo), mask.ctypes.data_as(C.c_void_p), P(q4), C.c_double(g), C.c_double(tol), 1)
    v, q, cov, ll = unpack(H, ns, st)
    assert rel(v, ob.vec[:ns]) < 1e-12 and rel(cov, ob.cov[:ns, :ns]) < 1e-12
    assert np.all(ll[mask == 0] == 0.0) and np.all(ll[mask == 1] != 0.0)


def six_row_block(ob, w, k, six, rng, B):
    """A six-row leg-odometry measurement near the current estimate, in pb_legodo_set_measurement_mode's layout:
    mode 1 (lin_rot_rate) z = (v, omega), idx 3,4,5,0,1,2; mode 2 (pos_and_lin_rate) z = (position, v), idx 9,10,11,3,4,5 with
    the per-filter three-row fall-back.  Returns lo12 [12][B], masks [2][B], idx."""
    idx = [3, 4, 5, 0, 1, 2] if six == 1 else [9, 10, 11, 3, 4, 5]
    z = ob.vec[idx, :] + 0.02 * rng.standard_normal((6, B))
    ra, rb = 0.01 + 0.02 * rng.random(B), 0.02 + 0.05 * rng.random(B)
    R = np.concatenate([np.tile(ra, (3, 1)), np.tile(rb, (3, 1))])
    masks = np.zeros((2, B), np.uint8)
    valid = (np.arange(B) + k) % 5 != 0
    full = valid & (((np.arange(B) + k) % 3 != 0) | (six == 1))
    masks[0] = full
    masks[1] = valid & ~full
    return np.ascontiguousarray(np.concatenate([z, R])), masks, idx


def apply_six_row_oracle(ob, six, lo12, masks, idx):
    ob.update_indexed(idx, lo12[0:6], lo12[6:12], mask=masks[0])
    if six == 2:
        ob.update_indexed([3, 4, 5], np.ascontiguousarray(lo12[3:6]), np.ascontiguousarray(lo12[9:12]), mask=masks[1])


@pytest.mark.parametrize("six", [1, 2])
@pytest.mark.parametrize("ns", [15, 21])
def test_cooperative_roles_with_six_row_leg_modes_match_oracle(oracle, harness, ns, six):
    """LegOdoCommon's lin_rot_rate / pos_and_lin_rate inside the step (SIX, rbis_coop.hpp): two 3-row blocks with ONE summed
    correction against the oracle's six-row update (and mode 2's three-row fall-back where the position is not valid)."""
    H = harness
    g, tol = oracle.constants()
    B, T = 24, 80
    w = Workload(B, n_states=ns)
    vec, quat, P0 = w.initial_state()
    P0 = P0 + random_spd(ns, B, 0.03, 9)
    if six == 1:
        # Every predict resets P(omega, omega) to q_gyro I and keeps the cross terms (rbis.cpp:121): with cross terms that large
        # next to q_gyro = 7.6e-5 the matrix is far from positive definite and S of a measurement that includes omega amplifies
        # rounding by many orders (the two statements of the same update then agree to 1e-6 only).  A filter only ever has the
        # cross terms an omega update left behind, which shrink by r / (q_gyro + r) per message.
        P0[0:3] *= 0.01
        P0[:, 0:3] *= 0.01
    if ns == 21:
        vec[15:18], vec[18:21] = 0.5 * w.bg, 0.5 * w.ba
    v21, P21 = embed21(vec, P0)
    ob = oracle.OracleBatch(v21, quat, P21)
    st = pack(H, ns, vec, quat, P0, np.zeros(B))
    q4 = w.process_noise()
    rng = np.random.default_rng(six)
    for k in range(T):
        imu = w.imu_block(k)
        ob.predict(imu, q4)
        lo12, masks, idx = six_row_block(ob, w, k, six, rng, B)
        apply_six_row_oracle(ob, six, lo12, masks, idx)
        H.hh_step_coop_six(ns, six, P(st), C.c_long(B), B, P(imu), P(lo12), masks.ctypes.data_as(C.c_void_p), P(q4),
                           C.c_double(g), C.c_double(tol))
    v, q, cov, ll = unpack(H, ns, st)
    assert rel(v, ob.vec[:ns]) < 1e-10 and rel(q, ob.quat) < 1e-10 and rel(cov, ob.cov[:ns, :ns]) < 1e-10
    assert rel(ll, ob.ll) < 1e-10


@pytest.mark.parametrize("six", [1, 2])
def test_four_wave_roles_with_six_row_leg_modes_match_oracle(oracle, harness, six):
    """The same on the four-wave mapping (SIX, rbis_quad.hpp): mode 1's angular-velocity block ahead of barrier A, mode 2's
    position block behind two more barriers."""
    H = harness
    g, tol = oracle.constants()
    ns, B, T = 21, 24, 80
    w = Workload(B, n_states=ns)
    vec, quat, P0 = w.initial_state()
    P0 = P0 + random_spd(ns, B, 0.03, 9)
    if six == 1:   # (see the two-wave test)
        P0[0:3] *= 0.01
        P0[:, 0:3] *= 0.01
    vec[15:18], vec[18:21] = 0.5 * w.bg, 0.5 * w.ba
    ob = oracle.OracleBatch(vec, quat, P0)
    st = pack(H, ns, vec, quat, P0, np.zeros(B))
    q4 = w.process_noise()
    rng = np.random.default_rng(six)
    for k in range(T):
        imu = w.imu_block(k)
        ob.predict(imu, q4)
        lo12, masks, idx = six_row_block(ob, w, k, six, rng, B)
        apply_six_row_oracle(ob, six, lo12, masks, idx)
        H.hh_step_quad_six(six, P(st), C.c_long(B), B, P(imu), P(lo12), masks.ctypes.data_as(C.c_void_p), P(q4),
                           C.c_double(g), C.c_double(tol))
    v, q, cov, ll = unpack(H, ns, st)
    assert rel(v, ob.vec[:ns]) < 1e-10 and rel(q, ob.quat) < 1e-10 and rel(cov, ob.cov[:ns, :ns]) < 1e-10
    assert rel(ll, ob.ll) < 1e-10

"""The ONE piece of this path that is pinned to the reference itself: TorqueAdjustment::processSample
(estimate_tools/src/backlash_filter_tools/torque_adjustment.cpp:27-62), compiled UNMODIFIED from /root/reference by `make -C oracle
ref` into oracle/_ref/libref_torque_adjustment.so (the image has no Eigen / boost / LCM / libbot2 / KDL: every other reference file
is unbuildable here, SURVEY.md 8c, and stays "parity unpinned").

Held to the reference's own object code, bit for bit (float arithmetic):
  * po_torque_adjust (oracle/leg_odometry.c),
  * oracle/leg_numpy.py's torque_adjust (the second witness),
  * the device function torque_adjust of rbis_legodo.hpp as compiled for the host harness,
  * the committed vectors ta_in / ta_out of tests/golden/leg_fk.npz,
  * the effort path of the golden leg-odometry fixtures: joint positions adjusted by the REFERENCE class, pushed through po_fk,
    give the fixtures' foot poses exactly.
CPU tier.  The .so travels with the repository snapshot (git-ignored, not gpurun-ignored); where neither it nor /root/reference
exists the tests skip."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref_torque_adjustment.so")
REF_SRC = "/root/reference/estimate_tools/src/backlash_filter_tools/torque_adjustment.cpp"


@pytest.fixture(scope="module")
def ref():
    if os.path.exists(REF_SRC):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "ref"])
    if not os.path.exists(REF_SO):
        pytest.skip("oracle/_ref is not built and /root/reference is not here")
    L = C.CDLL(REF_SO)
    L.ref_torque_adjustment.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_char_p),
                                        C.POINTER(C.c_float), C.POINTER(C.c_float)]
    return L


def ref_adjust(L, names, positions, efforts, adjust_names, gains):
    """One TorqueAdjustment(adjust_names, gains).processSample(names, positions, efforts) of the reference -> adjusted positions."""
    n, na = len(names), len(adjust_names)
    nm = (C.c_char_p * n)(*[s.encode() for s in names])
    an = (C.c_char_p * na)(*[s.encode() for s in adjust_names])
    pos = np.ascontiguousarray(positions, dtype=np.float32).copy()
    eff = np.ascontiguousarray(efforts, dtype=np.float32)
    g = np.ascontiguousarray(gains, dtype=np.float32)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    assert L.ref_torque_adjustment(na, an, fp(g), n, nm, fp(pos), fp(eff)) == 0
    return pos


def cases(n=4000, seed=11):
    rng = np.random.default_rng(seed)
    p = rng.normal(0, 1.0, n).astype(np.float32)
    e = (rng.normal(0, 300.0, n) * rng.choice([1.0, 10.0, 0.01], n)).astype(np.float32)
    g = rng.choice(np.array([7000, 10000, 1000, 50, 1e-3, 0.0, -0.0, np.inf, -np.inf, np.nan, 1e-40, -2500, 3e38], dtype=np.float32), n)
    # the clamp edges: effort / gain exactly at, just below and just above +-0.1
    p[:6] = 0.25
    g[:6] = 1000.0
    e[:6] = np.array([100.0, np.nextafter(np.float32(100.0), np.float32(0)), np.nextafter(np.float32(100.0), np.float32(200)),
                      -100.0, np.nextafter(np.float32(-100.0), np.float32(0)), np.nextafter(np.float32(-100.0), np.float32(-200))], dtype=np.float32)
    return p, e, g


def one_by_one(L, p, e, g):
    """every case as its own single-joint robot through the reference class (stdout of the constructors silenced)"""
    out = np.empty_like(p)
    fd = os.dup(1)
    null = os.open(os.devnull, os.O_WRONLY)
    os.dup2(null, 1)
    try:
        for i in range(len(p)):
            out[i] = ref_adjust(L, ["j"], p[i:i + 1], e[i:i + 1], ["j"], g[i:i + 1])[0]
    finally:
        os.dup2(fd, 1)
        os.close(fd)
        os.close(null)
    return out


def test_oracle_witness_and_device_function_equal_the_reference_bit_for_bit(ref, oracle, harness):
    from oracle import leg_numpy as ln
    L = oracle.lib()
    L.po_torque_adjust.restype = C.c_float
    L.po_torque_adjust.argtypes = [C.c_float, C.c_float, C.c_float]
    harness.hh_torque_adjust.restype = C.c_float
    harness.hh_torque_adjust.argtypes = [C.c_float, C.c_float, C.c_float]
    p, e, g = cases()
    want = one_by_one(ref, p, e, g)
    po = np.array([L.po_torque_adjust(a, b, c) for a, b, c in zip(p, e, g)], dtype=np.float32)
    # the device function is branch-free on gain == 0; the chain table stores 0 for a gain that is not std::isnormal
    # (leg_chain_set, rbis_legodo.hpp:197 = torque_adjustment.cpp:52) -- the same mapping here
    tiny = np.finfo(np.float32).tiny
    g_tab = np.where(np.isfinite(g) & (np.abs(g) >= tiny), g, np.float32(0.0)).astype(np.float32)
    hh = np.array([harness.hh_torque_adjust(a, b, c) for a, b, c in zip(p, e, g_tab)], dtype=np.float32)
    wit = np.array([ln.torque_adjust(a, b, c) for a, b, c in zip(p, e, g)], dtype=np.float32)
    assert po.tobytes() == want.tobytes(), np.nonzero(po.view(np.uint32) != want.view(np.uint32))[0][:10]
    assert hh.tobytes() == want.tobytes(), np.nonzero(hh.view(np.uint32) != want.view(np.uint32))[0][:10]
    assert wit.tobytes() == want.tobytes(), np.nonzero(wit.view(np.uint32) != want.view(np.uint32))[0][:10]
    assert (want != p).sum() > len(p) // 3 and (want == p).sum() > len(p) // 10    # both branches of isnormal(gain) were taken


def test_committed_vectors_are_the_reference_outputs(ref):
    g = np.load(os.path.join(ROOT, "tests", "golden", "leg_fk.npz"))
    p, e, gn = g["ta_in"]
    assert one_by_one(ref, p, e, gn).tobytes() == g["ta_out"].tobytes()


@pytest.mark.parametrize("name", ["legodo_alt", "legodo_standing", "legodo_ctrl"])
def test_effort_path_of_the_leg_fixtures_is_the_reference_class(ref, oracle, name):
    """Every robot of every tick of the fixture: ONE processSample call of the reference class on the whole 16-joint message with
    the fixture's adjustment joints / gains (joints looked up by NAME, as the reference does), then po_fk on the adjusted chain
    angles -> the fixture's foot poses, exactly."""
    import legs
    g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    L = oracle.lib()
    chain = legs.chain_arrays(legs.ATLAS_LEFT, legs.ATLAS_RIGHT, legs.ATLAS_ROWS)
    nl, nr, ty, rows, org, ax = chain
    names = ["other_%d" % r for r in range(legs.N_ROWS)]
    chain_names = [j[0] for j in legs.ATLAS_LEFT + legs.ATLAS_RIGHT]
    for j, r in enumerate(rows):
        names[r] = chain_names[j]
    T, _, B = g["jpos"].shape
    fd = os.dup(1)
    null = os.open(os.devnull, os.O_WRONLY)
    os.dup2(null, 1)
    try:
        worst = 0.0
        for k in range(0, T, 7):
            for b in range(B):
                adj = ref_adjust(ref, names, g["jpos"][k][:, b], g["jeff"][k][:, b], chain_names, g["gain"])
                for side, (lo, n) in enumerate(((0, nl), (nl, nr))):
                    ang = np.array([float(adj[rows[lo + j]]) for j in range(n)])
                    t, q = legs.oracle_fk(L, chain, side, ang)
                    worst = max(worst, float(np.max(np.abs(t - g["feet"][k][7 * side:7 * side + 3, b]))),
                                float(np.max(np.abs(q - g["feet"][k][7 * side + 3:7 * side + 7, b]))))
    finally:
        os.dup2(fd, 1)
        os.close(fd)
        os.close(null)
    assert worst == 0.0, worst

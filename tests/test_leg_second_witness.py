"""The leg-odometry oracle (oracle/leg_odometry.c: forward kinematics as KDL computes it, both foot-contact classifiers, the
walking-phase classifier, the foot-fixed pelvis integration, the world constraint, the torque adjustment) against a SECOND,
independent statement of the same reference code in numpy / scipy (oracle/leg_numpy.py: 4 x 4 homogeneous transforms,
scipy.spatial.transform.Rotation for every rotation): statuses bit-identical, poses <= 1e-12.  A change to either statement
that the other does not share fails here, on the CPU tier.  Reference: leg_estimate.cpp:147-556, FootContactAlt.cpp:35-130,
FootContact.cpp:29-83, foot_contact_classify.cpp:57-318, SignalTap.cpp:83-130, torque_adjustment.cpp:27-62."""
import ctypes as C

import numpy as np
import pytest

import legs
from oracle import leg_numpy as ln
from test_leg_odometry import OracleLegs, SCHMITT, STANDING, controller_contacts, gait


def _chains():
    return {"atlas": legs.chain_arrays(legs.ATLAS_LEFT, legs.ATLAS_RIGHT, legs.ATLAS_ROWS),
            "odd": legs.chain_arrays(legs.ODD_LEFT, legs.ODD_RIGHT, legs.ODD_ROWS)}


@pytest.mark.parametrize("which", ["atlas", "odd"])
def test_forward_kinematics_two_statements_agree(oracle, which):
    """po_fk (3 x 3 matrices, Rodrigues about the rotated axis, KDL's quaternion branches) against scipy: URDF fixed-axis rpy,
    Rotation.from_rotvec about the joint axis, 4 x 4 products from the root; revolute, fixed and prismatic joints, origins with
    rotations, skew and negative axes; angles over several turns."""
    L = oracle.lib()
    chain = _chains()[which]
    nl, nr, ty, rows, org, ax = chain
    rng = np.random.default_rng(77)
    worst_t = worst_r = 0.0
    for trial in range(300):
        for side, (lo, n) in enumerate(((0, nl), (nl, nr))):
            ang = rng.uniform(-7.0, 7.0, n) if trial % 3 else rng.uniform(-0.5, 0.5, n)
            t, q = legs.oracle_fk(L, chain, side, ang)
            T = ln.fk(ty[lo:lo + n], org[lo:lo + n], ax[lo:lo + n], ang)
            worst_t = max(worst_t, float(np.max(np.abs(T[:3, 3] - t))))
            worst_r = max(worst_r, float(np.max(np.abs(ln.from_tq(t, q)[:3, :3] - T[:3, :3]))))
            assert abs(np.linalg.norm(q) - 1.0) < 1e-14
    assert worst_t < 1e-12 and worst_r < 1e-12, (worst_t, worst_r)


def test_torque_adjustment_two_statements_agree(oracle):
    L = oracle.lib()
    L.po_torque_adjust.restype = C.c_float
    L.po_torque_adjust.argtypes = [C.c_float, C.c_float, C.c_float]
    rng = np.random.default_rng(5)
    gains = [7000.0, 10000.0, 0.0, float("inf"), float("nan"), 1e-40, -5000.0, 35.0]
    for _ in range(2000):
        p, e, g = np.float32(rng.normal()), np.float32(200 * rng.normal()), np.float32(gains[rng.integers(len(gains))])
        a = np.float32(L.po_torque_adjust(p, e, g))
        b = ln.torque_adjust(p, e, g)
        assert a.tobytes() == np.float32(b).tobytes(), (p, e, g, a, b)


@pytest.mark.parametrize("mode,fce", [("alt", True), ("alt", False), ("standing", True), ("ctrl", True)])
def test_leg_estimate_two_statements_agree(oracle, mode, fce):
    """leg_estimate::updateOdometry behind the forward kinematics, tick by tick on the same walking gait (with one 45 ms gap:
    the 30 ms reset): status, contact mode, primary foot, increment, pelvis pose and the world constraint of
    getLegOdometryWorldConstraint -- FootContactAlt, the "standing" FootContact classifier (float arithmetic) and the
    controller-contact override."""
    B, T = 5, 700
    standing = STANDING if mode == "standing" else None
    orc = OracleLegs(oracle, B, fce, standing, mode == "ctrl")
    wit = [ln.LegEstimate(*SCHMITT, fce, standing, mode == "ctrl") for _ in range(B)]
    rng = np.random.default_rng(3)
    seen = set()
    n_valid = n_pos = 0
    worst = dict(dt=0.0, dq=0.0, pos=0.0)
    for k, (utime, feet, forces, wq) in enumerate(gait(B, T, seed=12, gap_at=400)):
        nc = controller_contacts(k) if mode == "ctrl" else (-1, -1)
        wpos = np.ascontiguousarray(0.5 * rng.normal(size=(3, B)))
        od, os_, op = orc.update(utime, feet, forces, wq, nc, wpos)
        for b in range(B):
            w = wit[b]
            w.set_pose_body(ln.from_tq(wpos[:, b], wq[:, b]))
            w.set_foot_sensing(forces[0, b], forces[1, b])
            w.set_control_contacts(*nc)
            st, delta = w.update_odometry(utime, ln.from_tq(feet[0:3, b], feet[3:7, b]), ln.from_tq(feet[7:10, b], feet[10:14, b]))
            assert st == os_[b], (k, b, st, os_[b])
            assert w.previous_utime == op[b]
            if st >= 0:
                n_valid += 1
                worst["dt"] = max(worst["dt"], float(np.max(np.abs(delta[:3, 3] - od[0:3, b]))))
                worst["dq"] = max(worst["dq"], float(np.max(np.abs(ln.from_tq(np.zeros(3), od[3:7, b])[:3, :3] - delta[:3, :3]))))
                assert bool(orc.pos_ok[b]) == w.world_to_body_constraint_init, (k, b)
                if w.world_to_body_constraint_init:
                    n_pos += 1
                    worst["pos"] = max(worst["pos"], float(np.max(np.abs(w.world_to_body_constraint[:3, 3] - orc.pos[:, b]))))
        seen.update(np.unique(os_).tolist())
    assert n_valid > B * T // 4 and n_pos > B * T // 8
    assert seen == ({-1.0, 0.0, 1.0} if fce else {-1.0, 0.0})
    assert worst["dt"] < 1e-12 and worst["dq"] < 1e-12 and worst["pos"] < 1e-11, worst
    for b in range(B):
        t, q, info = orc.get(b)
        w = wit[b]
        assert info[0] == w.primary_foot and bool(info[1]) == w.leg_odo_init and info[2] == w.classify.mode and info[3] == w.classify.unknown
        assert np.max(np.abs(w.odom_to_body[:3, 3] - t)) < 1e-11
        assert np.max(np.abs(ln.from_tq(t, q)[:3, :3] - w.odom_to_body[:3, :3])) < 1e-12

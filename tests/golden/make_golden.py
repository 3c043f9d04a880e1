"""Generates the golden trajectory fixtures from the CPU oracle (oracle/pronto_oracle.c).

The reference itself ships no vectors for this path and cannot be built here, so these
fixtures pin the ORACLE (and, through it, the HIP path) across machines and rounds -- "parity unpinned"
still applies to the oracle-vs-reference link.  Inputs are not stored: they are regenerated from
pronto_amd/synth.py (counter-based, seed in the file).  Run: python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import po  # noqa: E402
from pronto_amd.synth import Workload  # noqa: E402
from util import embed21, run_config  # noqa: E402

CASES = {
    # name: (n_states, B, T, stride, vo_every, sm_every)   -- BASELINE.json configs 2, 3, 5 in miniature
    "n15_legodo": (15, 64, 1000, 100, 0, 0),
    "n15_legodo_vo": (15, 64, 1000, 100, 32, 0),
    "n21_legodo_scanmatch": (21, 64, 1000, 100, 0, 25),
    "n15_long": (15, 8, 10000, 1000, 0, 0),
}


def generate(name):
    n, B, T, stride, vo, sm = CASES[name]
    w = Workload(B, n_states=n)
    vec, quat, P0 = w.initial_state()
    v21, P21 = embed21(vec, P0)
    ob = po.OracleBatch(v21, quat, P21)
    snaps = dict(vec=[], quat=[], pdiag=[], ll=[], pvv=[])
    for s in range(T // stride):
        run_config(ob, w, stride, vo_every=vo, sm_every=sm, k0=s * stride)
        snaps["vec"].append(ob.vec.copy())
        snaps["quat"].append(ob.quat.copy())
        snaps["pdiag"].append(np.stack([ob.cov[i, i] for i in range(21)]))
        snaps["pvv"].append(ob.cov[3:6, 6:12].copy())  # a few off-diagonal blocks: [v, chi|pos]
        snaps["ll"].append(ob.ll.copy())
    out = {k: np.stack(v) for k, v in snaps.items()}
    g, tol = po.constants()
    out["meta"] = np.array([n, B, T, stride, vo, sm, g, tol])
    return out


SMOOTHER_CASES = {"smoother_n15": (15, 8, 24, 1e-3), "smoother_n21": (21, 8, 24, 1e-3)}  # n, B, T, dt


def generate_smoother(name):
    """Smoothed posterior at the first and the middle step of a T-step forward pass + backward recursion."""
    from smoother_ref import oracle_backward_pass
    n, B, T, dt = SMOOTHER_CASES[name]
    w = Workload(B, n_states=n)
    res = oracle_backward_pass(po, w, n, T, B, dt, keep=(0, T // 2))
    g, tol = po.constants()
    out = {"meta": np.array([n, B, T, dt, g, tol])}
    for tag, k in (("first", 0), ("mid", T // 2)):
        out["vec_" + tag], out["quat_" + tag], out["cov_" + tag] = res[k]
    return out


# ---- leg odometry, forward kinematics, joint filters, notch cascade (round 4) ------------------------------------------
# These fixtures carry their INPUTS as well (rounded to float32-representable doubles so that they compress): the GPU test
# that consumes them calls neither the oracle nor a generator.
LEG_CASES = {
    # name: (contact mode, filter_contact_events, world constraint exercised through wpos)
    "legodo_alt": ("alt", True),
    "legodo_alt_raw": ("alt", False),
    "legodo_standing": ("standing", True),
    "legodo_ctrl": ("ctrl", True),
}
LEG_B, LEG_T = 3, 420


def _f32r(a):
    return np.asarray(a, dtype=np.float32).astype(np.float64)


def generate_leg(name):
    """leg_estimate::updateOdometry from a joint state: per tick the two body-to-foot transforms (torque adjustment + forward
    kinematics), status, increment, previous utime and the world constraint (position + validity) of every robot; inputs: joint
    positions / efforts / foot forces (float32 like bot_core::joint_state_t), head pose, controller contact counts."""
    import legs
    import test_leg_odometry as tl
    mode, fce = LEG_CASES[name]
    standing = tl.STANDING if mode == "standing" else None
    orc = tl.OracleLegs(po, LEG_B, fce, standing, mode == "ctrl")
    chain = legs.chain_arrays(legs.ATLAS_LEFT, legs.ATLAS_RIGHT, legs.ATLAS_ROWS)
    gain = np.array([7000, 10000, 10000, 0, 10000, 10000] * 2, dtype=np.float32)
    rng = np.random.default_rng(41)
    ins = dict(utime=[], jpos=[], jeff=[], forces=[], wq=[], wpos=[], nc=[])
    outs = dict(feet=[], status=[], delta=[], prev=[], pos=[], pos_ok=[])
    for k, (utime, jp, je, forces, wq) in enumerate(legs.joint_gait(LEG_B, LEG_T, seed=31, dt_us=4000, gap_at=300)):
        wq = _f32r(wq)
        wq /= np.linalg.norm(wq, axis=0)                   # unit quaternions again after the rounding
        wpos = _f32r(0.5 * rng.normal(size=(3, LEG_B)))
        nc = tl.controller_contacts(k) if mode == "ctrl" else (-1, -1)
        feet = legs.oracle_feet(po.lib(), chain, jp, je, gain)
        od, os_, op = orc.update(utime, np.ascontiguousarray(feet), forces.astype(np.float64), np.ascontiguousarray(wq), nc,
                                 np.ascontiguousarray(wpos))
        for key, v in (("utime", utime), ("jpos", jp), ("jeff", je), ("forces", forces), ("wq", wq), ("wpos", wpos), ("nc", nc)):
            ins[key].append(np.array(v))
        for key, v in (("feet", feet), ("status", os_), ("delta", od), ("prev", op), ("pos", orc.pos), ("pos_ok", orc.pos_ok)):
            outs[key].append(np.array(v))
    out = {k: np.stack(v) for k, v in {**ins, **outs}.items()}
    out["status"] = out["status"].astype(np.int8)
    out["gain"] = gain
    final = [orc.get(b) for b in range(LEG_B)]
    out["final_body_t"] = np.stack([f[0] for f in final])
    out["final_body_q"] = np.stack([f[1] for f in final])
    out["final_info"] = np.array([f[2] for f in final])
    out["meta"] = np.array([tl.SCHMITT[0], tl.SCHMITT[1], tl.SCHMITT[2], tl.SCHMITT[3], float(fce), float(standing is not None),
                            (standing or (0.0, 0.0))[0], (standing or (0.0, 0.0))[1], float(mode == "ctrl")])
    return out


def generate_fk():
    """po_fk on both test chains (tests/legs.py): 48 angle sets each -> body-to-foot (t, q) of the two legs; and the torque
    adjustment on a grid of (position, effort, gain) floats."""
    import ctypes as C
    import legs
    L = po.lib()
    L.po_torque_adjust.restype = C.c_float
    L.po_torque_adjust.argtypes = [C.c_float, C.c_float, C.c_float]
    rng = np.random.default_rng(43)
    out = {}
    for tag, chain in (("atlas", legs.chain_arrays(legs.ATLAS_LEFT, legs.ATLAS_RIGHT, legs.ATLAS_ROWS)),
                       ("odd", legs.chain_arrays(legs.ODD_LEFT, legs.ODD_RIGHT, legs.ODD_ROWS))):
        nl, nr, ty, rows, org, ax = chain
        jp = _f32r(rng.uniform(-2.5, 2.5, size=(legs.N_ROWS, 48))).astype(np.float32)
        out[tag + "_jpos"] = jp
        out[tag + "_feet"] = legs.oracle_feet(L, chain, jp)
    p, e = _f32r(rng.normal(size=256)).astype(np.float32), _f32r(200 * rng.normal(size=256)).astype(np.float32)
    g = np.array([7000.0, 10000.0, 0.0, np.inf, 35.0, -5000.0, np.nan, 1e-40] * 32, dtype=np.float32)
    out["ta_in"] = np.stack([p, e, g])
    out["ta_out"] = np.array([L.po_torque_adjust(a, b, c) for a, b, c in zip(p, e, g)], dtype=np.float32)
    return out


def generate_joint_filter():
    """leg_estimate.cpp:411-428 on 28 + 2 joint rows of 3 robots over 60 messages with uneven time stamps: the low-pass and
    the Kalman outputs as float32 (the reference's std::vector<float>), to be matched to the BIT."""
    from test_joint_filter import OracleJointFilter
    rng = np.random.default_rng(47)
    B, T, rows = 3, 60, 30
    ts = 1_000_000 + np.cumsum(rng.choice([2000, 2000, 3000, 1000], size=T))
    t = ts * 1e-6
    jp = (0.4 * np.sin(3 * t)[:, None, None] * rng.uniform(0.5, 1.5, size=(1, rows, B)) + 0.002 * rng.normal(size=(T, rows, B))).astype(np.float32)
    jv = (1.2 * np.cos(3 * t)[:, None, None] * np.ones((1, rows, B))).astype(np.float32)
    out = dict(utime=ts.astype(np.int64), jpos=jp, jvel=jv, noise=np.array([0.01, 5e-4, 5e-4]))
    for mode in ("lowpass", "kalman"):
        f = OracleJointFilter(po.lib(), B, mode, noise=(0.01, 5e-4, 5e-4))
        out[mode] = np.stack([f.apply(int(ts[k]), jp[k], jv[k]) for k in range(T)])
    return out


def generate_notch():
    """InsHandler::doFilter (sensor_handlers.cpp:154-162): 300 packets of 3 robots through the 87 / 174 / 348 Hz cascade."""
    rng = np.random.default_rng(53)
    T, B = 300, 3
    t = np.arange(T) * 1e-3
    acc = _f32r(9.8 * (np.arange(3) == 2)[None, :, None] + rng.normal(size=(T, 3, B)) + np.sin(2 * np.pi * 87 * t)[:, None, None])
    filt = np.stack([po.notch_cascade_run(np.ascontiguousarray(acc[:, :, b]), 87.0) for b in range(B)], axis=2)
    return dict(acc=acc, filtered=filt, meta=np.array([87.0, 1000.0]))


LEG_FIXTURES = {**{n: (lambda n=n: generate_leg(n)) for n in LEG_CASES}, "leg_fk": generate_fk, "joint_filter": generate_joint_filter,
                "imu_notch": generate_notch}


if __name__ == "__main__":
    only = sys.argv[1:]
    for name, gen in LEG_FIXTURES.items():
        if only and name not in only and "legs" not in only:
            continue
        out = gen()
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, {k: v.shape for k, v in out.items()})
    if only:
        sys.exit(0)
    for name in SMOOTHER_CASES:
        out = generate_smoother(name)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, {k: v.shape for k, v in out.items()})
    for name in CASES:
        out = generate(name)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, {k: v.shape for k, v in out.items()})

"""Rates of the leg-odometry kernels at 64k filters with PER-FILTER device-resident inputs (independent robots / log
segments; scripts/shim_rate.sh measures the one-log-for-every-filter sweep through the handler API): the odometry alone
(k_legodo, from foot poses and from joint states = + forward kinematics), and the whole IMU + joint-state pair as one call
(pb_step_legodo_joints / _feet: one kernel for 15 states).  Wall clock around back-to-back launches."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import legs  # noqa: E402
from pronto_amd.batch import BatchEstimator  # noqa: E402
from pronto_amd.synth import Workload  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
PAIRS_ONLY = len(sys.argv) > 2 and sys.argv[2] == "pairs"   # A/B runs of the pair kernels: skip the other rows


def timeit(fn, reps=400):
    for _ in range(100):  # (long enough for the clocks to settle: with 6 + 60 launches the first rows of a run read 20-30 % high)
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


chain = legs.chain_arrays(legs.ATLAS_LEFT, legs.ATLAS_RIGHT, list(range(12)))
gain = np.array([7000, 10000, 10000, 10000, 10000, 10000] * 2, dtype=np.float32)
msgs = legs.joint_gait(B, 8, seed=1, n_rows=12, rows=list(range(12)))
import test_leg_odometry as tl  # noqa: E402
fmsgs = tl.gait(B, 8, seed=1)
up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
for n in (15, 21):
    w = Workload(B, n_states=n, dt_us=2000)
    vec, quat, P0 = w.initial_state()
    est = BatchEstimator(B, n_states=n)
    est.reset(vec, quat, P0)
    est.legodo_init(475.0, 525.0, 7000, 7000, True)
    est.legodo_set_chain(*chain, gain)
    q4 = w.process_noise()
    imu = up(w.imu_block(0))
    d_lo = torch.zeros((6, B), dtype=torch.float64, device=dev)
    d_mask = torch.zeros(B, dtype=torch.uint8, device=dev)
    jm = [(m[0], up(m[1]), up(m[2]), up(m[3])) for m in msgs]
    fm = [(m[0], up(m[1]), up(m[2])) for m in fmsgs]
    st = (n + 5 + n * (n + 1) // 2) * 8
    leg = 2 * 136
    k = [0]

    def nxt(lst):
        k[0] += 1
        return lst[k[0] % len(lst)]

    rows = []
    if not PAIRS_ONLY:
        t = timeit(lambda: est.legodo_update(*nxt(fm), 0.1, 0.5, None, None, d_lo, d_mask, after_predict=imu))
        rows.append(("k_legodo: foot poses -> measurement (after_predict)", t, leg + (n + 4) * 8 + 128 + 56 + 49))
        t = timeit(lambda: (lambda m: est.legodo_update_joints(m[0], m[1], None, m[3], 0.1, 0.5, None, None, d_lo, d_mask, after_predict=imu))(nxt(jm)))
        rows.append(("k_legodo: joint state -> FK -> measurement", t, leg + (n + 4) * 8 + 56 + 56 + 49))
        t = timeit(lambda: (lambda m: est.legodo_update_joints(m[0], m[1], m[2], m[3], 0.1, 0.5, None, None, d_lo, d_mask, after_predict=imu))(nxt(jm)))
        rows.append(("k_legodo: joint state + efforts (torque adjustment)", t, leg + (n + 4) * 8 + 104 + 56 + 49))
        t = timeit(lambda: est.step_legodo(imu, d_lo, d_mask, q4))
        rows.append(("fused step reading that measurement", t, 2 * st + 104))
    t = timeit(lambda: (lambda m: est.step_legodo_feet(imu, q4, m[0], m[1], m[2], 0.1, 0.5))(nxt(fm)))
    rows.append(("pair in one call: IMU + foot poses", t, 2 * st + 56 + leg + 128))
    t = timeit(lambda: (lambda m: est.step_legodo_joints(imu, q4, m[0], m[1], None, m[3], 0.1, 0.5))(nxt(jm)))
    rows.append(("pair in one call: IMU + joint state", t, 2 * st + 56 + leg + 56))
    t = timeit(lambda: (lambda m: est.step_legodo_joints(imu, q4, m[0], m[1], m[2], m[3], 0.1, 0.5))(nxt(jm)))
    rows.append(("pair in one call: IMU + joint state + efforts", t, 2 * st + 56 + leg + 104))
    # LegOdoCommon's six-row modes in the same kernel (SIX, rbis_legstep.hpp); the lin_rate rows above are the yardstick
    for mode, name in ((1, "lin_rot_rate"), (2, "pos_and_lin_rate")):
        est.legodo_set_measurement_mode(mode, 0.05, 0.4, 0.9)
        t = timeit(lambda: (lambda m: est.step_legodo_feet(imu, q4, m[0], m[1], m[2], 0.1, 0.5))(nxt(fm)))
        rows.append(("pair, mode %s: IMU + foot poses" % name, t, 2 * st + 56 + leg + 128 + (48 if mode == 2 else 0)))
        t = timeit(lambda: (lambda m: est.step_legodo_joints(imu, q4, m[0], m[1], m[2], m[3], 0.1, 0.5))(nxt(jm)))
        rows.append(("pair, mode %s: IMU + joint state + efforts" % name, t, 2 * st + 56 + leg + 104 + (48 if mode == 2 else 0)))
    est.legodo_set_measurement_mode(0)
    if n == 15 and not PAIRS_ONLY:
        # the joint filters in front of the kinematics (pb_joint_filter): 12 chain rows; low-pass = 13 window floats in + 1 out,
        # Kalman = 6 doubles in + 6 out, + the float in / out of each row
        d_f = torch.zeros((12, B), dtype=torch.float32, device=dev)
        jv = up(np.zeros((12, B), dtype=np.float32))
        for mode, nb in (("lowpass", 12 * (13 * 4 + 4 + 8)), ("kalman", 12 * (96 + 8))):
            est.joint_filter_init(mode, 0.01, 5e-4, 5e-4)
            ut = [1_000_000]

            def one():
                ut[0] += 2000
                m = nxt(jm)
                est.joint_filter(ut[0], m[1], jv, m[2], d_f)
            t = timeit(one)
            rows.append(("k_joint_filter %s: 12 joints (+ torque adjustment)" % mode, t, nb))
    for name, t, nb in rows:
        print("n=%d B=%d %-52s %7.1f us  %6.0f GB/s  frac %.3f" % (n, B, name, t * 1e6, nb * B / t / 1e9, nb * B / t / 1e9 / 8000))
    est.close()

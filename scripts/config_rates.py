"""Whole-configuration throughput for BASELINE.json configs 3 and 5 (the fused step plus the sparse corrections), device
resident inputs, wall clock around the launches:
  config 3: 64k filters, n = 15, predict + legodo every step, VO position_orient (m = 6: snapshot -> compose -> update ->
            new keyframe) every 32nd step;
  config 5: 64k filters, n = 21, predict + legodo every step, scan-match position_yaw (m = 4) every 25th step.
Steps counted = IMU ticks x filters; bytes = the T = 1 accounting plus the extras on their steps."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pronto_amd.batch import BatchEstimator  # noqa: E402
from pronto_amd.synth import Workload  # noqa: E402

dev = torch.device("cuda:0")
B, T = 65536, 800
up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)

from pronto_amd._lib import PB_CORR_POS_ORIENT, PB_CORR_POS_YAW  # noqa: E402

# "fused": on a correction tick the three updates run as ONE kernel (pb_step_legodo_correct) instead of the fused step
# followed by the generic update kernel
for name, n, vo, sm, fused in (("config 3 (n=15, VO every 32nd), three launches", 15, 32, 0, False),
                               ("config 3 (n=15, VO every 32nd), ONE launch on VO ticks", 15, 32, 0, True),
                               ("config 5 (n=21, scan-match every 25th), two launches", 21, 0, 25, False),
                               ("config 5 (n=21, scan-match every 25th), ONE launch on scan-match ticks", 21, 0, 25, True),
                               ("config 2 (n=15, legodo only)", 15, 0, 0, False)):
    w = Workload(B, n_states=n)
    vec, quat, P0 = w.initial_state()
    est = BatchEstimator(B, n_states=n, n_snapshots=1)
    est.reset(vec, quat, P0)
    est.snapshot(0)
    q4 = w.process_noise()
    K = 16  # distinct input blocks, cycled (generation is host work, not what is measured)
    imu = [up(w.imu_block(k)) for k in range(K)]
    los = [w.legodo_block(k) for k in range(K)]
    lo, mask = [up(a) for a, _ in los], [up(m) for _, m in los]
    z, qm, Rd = w.vo_block(0)
    t_delta, q_delta, d_Rd = up(0.01 * z), up(qm), up(Rd)
    z6 = torch.zeros((6, B), dtype=torch.float64, device=dev)
    q_out = torch.empty((4, B), dtype=torch.float64, device=dev)
    zs, qs, Rs = w.scanmatch_block(0)
    z4, d_qs, d_Rs = up(np.vstack([zs, np.zeros((1, B))])), up(qs), up(Rs)

    def run(steps):
        for k in range(steps):
            is_vo, is_sm = vo and k % vo == vo - 1, sm and k % sm == sm - 1
            if fused and is_vo:
                est.compose_delta(0, t_delta, q_delta, z6[0:3], q_out)
                est.step_legodo_correct(imu[k % K], lo[k % K], mask[k % K], q4, PB_CORR_POS_ORIENT, z6, d_Rd, q_out)
                est.snapshot(0)
                continue
            if fused and is_sm:
                est.step_legodo_correct(imu[k % K], lo[k % K], mask[k % K], q4, PB_CORR_POS_YAW, z4, d_Rs, d_qs)
                continue
            est.step_legodo(imu[k % K], lo[k % K], mask[k % K], q4)
            if is_vo:
                est.compose_delta(0, t_delta, q_delta, z6[0:3], q_out)
                est.update_indexed([9, 10, 11, 6, 7, 8], z6, d_Rd, quat_meas=q_out)
                est.snapshot(0)
            if is_sm:
                est.update_indexed([9, 10, 11, 8], z4, d_Rs, quat_meas=d_qs)
    run(64)
    est.sync()
    t0 = time.perf_counter()
    run(T)
    est.sync()
    dt = time.perf_counter() - t0
    st = (n + 5 + n * (n + 1) // 2) * 8
    # algorithmic bytes: the T = 1 step plus, on correction ticks, the measurement (+ the snapshot / compose traffic) and --
    # only when the correction is a launch of its own -- one more state round trip
    extra_rt = 0 if fused else 2 * st
    bytes_step = 2 * st + 104 + ((extra_rt + 104 + 56 + 56) / vo if vo else 0) + ((extra_rt + 88) / sm if sm else 0)
    print("%-72s %.3e steps/s, %6.2f us per tick, %.0f GB/s algorithmic = %.2f of the HBM roofline"
          % (name, B * T / dt, dt / T * 1e6, bytes_step * B * T / dt / 1e9, bytes_step * B * T / dt / 1e9 / 8000))
    est.close()

"""Per-kernel rates at BASELINE batch size (64k filters) for the kernels next to the fused step: predict only,
generic indexed / indexed+orientation updates (VO m=6, scan-match m=4, legodo m=3), snapshot/compose, notch.
Wall clock around back-to-back launches on device-resident inputs; algorithmic bytes = one state round trip + inputs."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pronto_amd.batch import BatchEstimator  # noqa: E402
from pronto_amd.synth import Workload  # noqa: E402

dev = torch.device("cuda:0")
B = 65536


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for n in (15, 21):
    w = Workload(B, n_states=n)
    vec, quat, P0 = w.initial_state()
    est = BatchEstimator(B, n_states=n)
    est.reset(vec, quat, P0)
    q4 = w.process_noise()
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    imu = up(w.imu_block(0))
    lo, mask = w.legodo_block(0)
    d_lo, d_mask = up(lo), up(mask)
    st = (n + 5 + n * (n + 1) // 2) * 8
    rows = []
    t = timeit(lambda: est.predict(imu, q4))
    rows.append(("predict (k_step/k_step_coop, UPDATE=false)", t, 2 * st + 56))
    t = timeit(lambda: est.step_legodo(imu, d_lo, d_mask, q4))
    rows.append(("fused step", t, 2 * st + 104))
    z3, r3 = up(lo[0:3]), up(lo[3:6])
    t = timeit(lambda: est.update_indexed([3, 4, 5], z3, r3, mask=d_mask))
    rows.append(("k_update m=3 (legodo, per-filter diag R, mask)", t, 2 * st + 48))
    z, qm, Rd = w.vo_block(0)
    z6 = up(np.vstack([z, np.zeros((3, B))]))
    d_qm, d_Rd = up(qm), up(Rd)
    t = timeit(lambda: est.update_indexed([9, 10, 11, 6, 7, 8], z6, d_Rd, quat_meas=d_qm))
    rows.append(("k_update m=6 orient (VO position_orient)", t, 2 * st + 104))
    z, qm, Rd = w.scanmatch_block(0)
    z4 = up(np.vstack([z, np.zeros((1, B))]))
    d_qm4, d_Rd4 = up(qm), up(Rd)
    t = timeit(lambda: est.update_indexed([9, 10, 11, 8], z4, d_Rd4, quat_meas=d_qm4))
    rows.append(("k_update m=4 orient (scan-match position_yaw)", t, 2 * st + 88))
    z6r, r6r = up(np.vstack([lo[0:3], 0.01 * np.ones((3, B))])), up(np.vstack([lo[3:6], 0.09 * np.ones((3, B))]))
    t = timeit(lambda: est.update_indexed([3, 4, 5, 0, 1, 2], z6r, r6r, mask=d_mask))
    rows.append(("m=6 legodo lin_rot_rate (15: k_update_lane, 21: k_update_quad_list)", t, 2 * st + 96))
    # free-form run-time index lists (pronto_indexed_measurement_t: any states): 15 states k_update_lane_rt (m <= 4) / k_update_coop_rt (m = 5, 6), 21 states k_update_quad_rt
    zf3, rf3 = up(0.1 * np.ones((3, B))), up(0.04 * np.ones((3, B)))
    t = timeit(lambda: est.update_indexed([2, 9, n - 1], zf3, rf3, mask=d_mask))
    rows.append(("free-form list m=3 [2, 9, n-1]", t, 2 * st + 48))
    zf6, rf6 = up(0.1 * np.ones((6, B))), up(0.04 * np.ones((6, B)))
    t = timeit(lambda: est.update_indexed([0, 4, 8, 12, n - 5, n - 1], zf6, rf6, mask=d_mask))
    rows.append(("free-form list m=6 [0, 4, 8, 12, n-5, n-1]", t, 2 * st + 96))
    for name, t, nb in rows:
        print("n=%d %-56s %7.1f us  %6.0f GB/s  frac %.3f" % (n, name, t * 1e6, nb * B / t / 1e9, nb * B / t / 1e9 / 8000))

# IMU front end: the 3-stage notch cascade (k_notch), 3 packets per message (the KVH batch's typical new-packet count),
# device-resident packets; bytes = packets in + the 36-double filter state read and written + the filtered sample out
est = BatchEstimator(B, n_states=15)
est.imu_notch_init(87.0, 1000.0)
pk = torch.randn((3, 3, B), dtype=torch.float64, device=dev)
ao = torch.empty((3, B), dtype=torch.float64, device=dev)
t = timeit(lambda: est.imu_notch(pk, ao))
nb = (3 * 3 + 2 * 36 + 3) * 8
print("k_notch, 3 packets per message                               %7.1f us  %6.0f GB/s  frac %.3f" % (t * 1e6, nb * B / t / 1e9, nb * B / t / 1e9 / 8000))
est.close()

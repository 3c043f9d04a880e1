"""Throughput of the RTS smoother step kernel (pb_smooth_step) at BASELINE batch size: 64k filters, wall clock around back-to-back launches + sync."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pronto_amd.batch import BatchEstimator  # noqa: E402
from pronto_amd.synth import Workload  # noqa: E402

for n in (15, 21):
    B = int(os.environ.get("SMOOTH_B", "65536"))
    w = Workload(B, n_states=n)
    vec, quat, P0 = w.initial_state()
    est = BatchEstimator(B, n_states=n)
    est.reset(vec, quat, P0)
    est.history_reserve(4)
    q4 = w.process_noise()
    est.state_save(0)
    est.predict(w.imu_block(0), q4)
    est.state_save(1)
    lo, mask = w.legodo_block(0)
    est.update_indexed([3, 4, 5], np.ascontiguousarray(lo[0:3]), np.ascontiguousarray(lo[3:6]), mask=mask)
    est.state_save(2)
    for _ in range(50):
        est.smooth_step(1, 2, 0, 3, 1e-3)
    est.sync()
    import time
    reps = 200  # (20 launches were 4 ms of work: too short for the clocks to settle)
    t0 = time.perf_counter()
    for _ in range(reps):
        est.smooth_step(1, 2, 0, 3, 1e-3)
    est.sync()
    ms = (time.perf_counter() - t0) / reps * 1e3
    nbytes = (3 * (n * (n + 1) // 2 + n + 4) + (n * (n + 1) // 2 + n + 5)) * 8
    print("smoother step (pb_smooth_step), n=%d: %d filters, %.1f us/step, %.3e filter-steps/s, %.0f GB/s algorithmic (%d B/filter)"
          % (n, B, ms * 1e3, B / (ms * 1e-3), nbytes * B / (ms * 1e-3) / 1e9, nbytes))

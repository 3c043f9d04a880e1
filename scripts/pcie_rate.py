"""PCIe-inclusive rate of the hot step when the boundary hands over HOST buffers (pb_step_legodo, PB_HOST):
each step stages 104 B/filter over PCIe before the launch.  Reported in DESIGN.md; never bench.py's `value`."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from pronto_amd.batch import BatchEstimator  # noqa: E402
from pronto_amd.synth import Workload  # noqa: E402

B, T = 65536, 60
w = Workload(B, n_states=15)
vec, quat, P0 = w.initial_state()
imu, lo, mask = w.streams(0, T)
q4 = w.process_noise()
est = BatchEstimator(B, n_states=15)
est.reset(vec, quat, P0)
for k in range(10):
    est.step_legodo(imu[k], lo[k], mask[k], q4)
est.sync()
t0 = time.perf_counter()
for k in range(10, T):
    est.step_legodo(imu[k], lo[k], mask[k], q4)
est.sync()
dt = time.perf_counter() - t0
print("PCIe-inclusive: %d filters x %d steps from pageable host blocks: %.3e steps/s, %.3f ms/step, %.1f GB/s over PCIe"
      % (B, T - 10, B * (T - 10) / dt, dt / (T - 10) * 1e3, 105 * B * (T - 10) / dt / 1e9))

# the same loop when ONE robot's stream feeds every filter (parameter sweep, LogPlayer): PB_HOST_BROADCAST blocks of
# [7] + [6] values that travel as kernel arguments -- no device block, no fill launch, no input traffic
est.reset(vec, quat, P0)
imu1 = np.ascontiguousarray(imu[:, :, 0])
lo1 = np.ascontiguousarray(lo[:, :, 0])
for k in range(10):
    est.step_legodo(imu1[k], lo1[k], None, q4)
est.sync()
t0 = time.perf_counter()
for k in range(10, T):
    est.step_legodo(imu1[k], lo1[k], None, q4)
est.sync()
dt = time.perf_counter() - t0
print("broadcast (one message for all filters): %d filters x %d steps: %.3e steps/s, %.3f ms/step"
      % (B, T - 10, B * (T - 10) / dt, dt / (T - 10) * 1e3))

# per-filter host blocks again, from page-locked buffers (pb_host_alloc): the staging copy is a DMA at link rate and
# overlaps the previous step's kernel
p_imu, p_lo, p_mask = est.pinned_empty((7, B)), est.pinned_empty((6, B)), est.pinned_empty((B,), np.uint8)
est.reset(vec, quat, P0)
for k in range(10):
    p_imu[:], p_lo[:], p_mask[:] = imu[k], lo[k], mask[k]
    est.step_legodo(p_imu, p_lo, p_mask, q4)
est.sync()
p_imu[:], p_lo[:], p_mask[:] = imu[10], lo[10], mask[10]
t0 = time.perf_counter()
for k in range(10, T):
    est.step_legodo(p_imu, p_lo, p_mask, q4)
est.sync()
dt = time.perf_counter() - t0
print("PCIe-inclusive from pinned host blocks: %.3e steps/s, %.3f ms/step, %.1f GB/s over PCIe"
      % (B * (T - 10) / dt, dt / (T - 10) * 1e3, 105 * B * (T - 10) / dt / 1e9))

"""Whole-log RTS smoothing with bounded memory (pb_smooth_log: checkpoint and recompute): device time per smoothed filter-step,
forward pass + recompute + smoother steps, and the slots it took against the 2 T a posterior per update would need.
  python scripts/smooth_log_rate.py            env: SMOOTH_LOG_CASES="n,B,T,K;..." """
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from pronto_amd.batch import BatchEstimator  # noqa: E402
from pronto_amd.synth_device import DeviceWorkload  # noqa: E402

cases = os.environ.get("SMOOTH_LOG_CASES", "15,4096,10000,64;21,4096,10000,64;15,65536,1000,22;21,65536,1000,22")
for case in cases.split(";"):
    n, B, T, K = (int(v) for v in case.split(","))
    dw = DeviceWorkload(B, n_states=n, device="cuda:0")
    vec, quat, P0 = dw.host.initial_state()
    q4 = dw.host.process_noise()
    imu, lo, mask = dw.streams(0, T)
    est = BatchEstimator(B, n_states=n)
    est.reset(vec, quat, P0)
    need = est.smooth_log_slots(T, K)
    est.history_reserve(need)
    est.smooth_log(imu[:2 * K], lo[:2 * K], mask[:2 * K], q4, 1e-3, K)   # warm-up
    est.reset(vec, quat, P0)
    ms = est.smooth_log(imu, lo, mask, q4, 1e-3, K, timed=True)
    per_slot = (n + 5 + n * (n + 1) // 2) * 8 * B / 1e6
    s = est.summary()
    print("pb_smooth_log n=%d: %d steps x %d filters, stride %d: %d slots (%.1f GB; a posterior per update: %d slots, %.0f GB): %.1f ms = %.2f us per "
          "smoothed step of the batch = %.4f us per smoothed filter-step; nonfinite %d"
          % (n, T, B, K, need, need * per_slot / 1e3, 2 * T, 2 * T * per_slot / 1e3, ms, ms * 1e3 / T, ms * 1e3 / (T * B), int(s[3])), flush=True)
    est.close()
    del imu, lo, mask

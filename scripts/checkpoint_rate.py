"""Forward pass that keeps EVERY posterior (what the reference's history does by value, mav_state_est.cpp:55-61, and
what the smoother needs): step + pb_state_save copy against pb_set_output_slot (the step writes into the slot)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pronto_amd.batch import BatchEstimator  # noqa: E402
from pronto_amd.synth import Workload  # noqa: E402

dev = torch.device("cuda:0")
B, T = 65536, 64
for n in (15, 21):
    w = Workload(B, n_states=n)
    vec, quat, P0 = w.initial_state()
    q4 = w.process_noise()
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    imu, (lo, mask) = up(w.imu_block(0)), w.legodo_block(0)
    lo, mask = up(lo), up(mask)
    est = BatchEstimator(B, n_states=n)
    est.reset(vec, quat, P0)
    est.history_reserve(T)
    res = {}
    for mode in ("copy", "slot", "none"):
        for rep in range(2):
            est.reset(vec, quat, P0)
            est.sync()
            t0 = time.perf_counter()
            for k in range(T):
                if mode == "slot":
                    est.set_output_slot(k)
                est.step_legodo(imu, lo, mask, q4)
                if mode == "copy":
                    est.state_save(k)
            est.sync()
            res[mode] = (time.perf_counter() - t0) / T
    # the write-through replay: T steps per launch, every posterior into its slot, nothing read back
    imus, los, masks = (a.unsqueeze(0).repeat(T, *([1] * a.dim())).contiguous() for a in (imu, lo, mask))
    wt = {}
    for Tf in (8, 32):
        for rep in range(2):
            est.reset(vec, quat, P0)
            est.sync()
            wt[Tf] = est.replay_legodo_checkpointed(imus, los, masks, q4, Tf, first_slot=0, timed=True) / T * 1e-3
    st = (n + 5 + n * (n + 1) // 2) * 8
    print("n=%d, 64k filters, per step: no checkpoints %.1f us | step + copy %.1f us | step into the slot %.1f us | write-through replay "
          "T=8 %.1f us, T=32 %.1f us (%.0f B per filter-step under its own accounting: %.2f of the HBM roofline)"
          % (n, res["none"] * 1e6, res["copy"] * 1e6, res["slot"] * 1e6, wt[8] * 1e6, wt[32] * 1e6, st + 104 + 2 * st / 32,
             (st + 104 + 2 * st / 32) * B / wt[32] / 8e12))
    est.close()

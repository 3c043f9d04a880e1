"""Fused-step time per launch over batch sizes and kernel variants (HIP events around back-to-back launches, inputs
resident in HBM, 16 distinct input blocks cycled; a 4096-filter synthetic workload tiled along the filter axis -- a
bandwidth measurement, parity is tested elsewhere).
  python scripts/batch_sweep.py [n_states ...]      env: SWEEP_SIZES="65536,1048576"  SWEEP_VARIANTS="default,coop0,..."
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from pronto_amd.batch import BatchEstimator  # noqa: E402
from pronto_amd.synth import Workload  # noqa: E402

dev = torch.device("cuda:0")
ns = [int(a) for a in sys.argv[1:]] or [15, 21]
sizes = [int(a) for a in os.environ.get("SWEEP_SIZES", "32768,65536,131072,196608,262144,524288,1048576").split(",")]
VARIANTS = {"default": {}, "coop0": {"PRONTO_BATCH_COOP15": "0"}, "coop1": {"PRONTO_BATCH_COOP15": "1"},
            "xcd0": {"PRONTO_BATCH_XCD": "0"}, "xcd1": {"PRONTO_BATCH_XCD": "1"},
            "quad0": {"PRONTO_BATCH_QUAD21": "0"}, "mh0": {"PRONTO_BATCH_MEMHINT": "0"}, "mh1": {"PRONTO_BATCH_MEMHINT": "1"}, "mh2": {"PRONTO_BATCH_MEMHINT": "2"},
            # launch order of run_legodo beyond the memory-side cache: "default" is the library's choice (blocked there), "unblocked"
            # the step-by-step order over the whole batch, "blk<N>k" a named block size
            "unblocked": {"PRONTO_BATCH_BLOCKED": "0"}, "blocked": {"PRONTO_BATCH_BLOCKED": "1"},
            # two workgroups per 64-filter tile (15 states, small batches): off / on
            "half0": {"PRONTO_BATCH_HALF": "0"}, "half1": {"PRONTO_BATCH_HALF": "1"}}
for _kb in (48, 64, 80, 96, 112, 128, 160, 192, 224, 256):
    VARIANTS["blk%dk" % _kb] = {"PRONTO_BATCH_BLOCKED": "1", "PRONTO_BATCH_BLOCK_FILTERS": str(_kb * 1024)}
variants = os.environ.get("SWEEP_VARIANTS", "default").split(",")
# SWEEP_STEPS: steps per run_legodo call (distinct input blocks cycled).  The cache-blocked order runs a block of filters through ALL of
# them before the next block starts, so its rate depends on the stream length: per block one cold read and one write-back of the
# block's state are shared by K steps (K = 16: +10-16 % on top of K warm steps; a real replay has thousands of steps per call).
K, B0 = int(os.environ.get("SWEEP_STEPS", "16")), 4096
for n in ns:
    w = Workload(B0, n_states=n)
    vec, quat, P0 = w.initial_state()
    q4 = w.process_noise()
    imu, lo, mask = w.streams(0, K)
    bps = 2 * ((n + 4) * 8 + n * (n + 1) // 2 * 8 + 8) + 104
    for B in sizes:
        if n == 21 and B > 524288:
            continue
        r = B // B0
        d_imu = torch.from_numpy(imu).to(dev).repeat(1, 1, r).contiguous()
        d_lo = torch.from_numpy(lo).to(dev).repeat(1, 1, r).contiguous()
        d_mask = torch.from_numpy(mask).to(dev).repeat(1, r).contiguous()
        tv = torch.from_numpy(vec).to(dev).repeat(1, r).contiguous()
        tq = torch.from_numpy(quat).to(dev).repeat(1, r).contiguous()
        tP = torch.from_numpy(P0).to(dev).repeat(1, 1, r).contiguous()
        for v in variants:
            for k_, x in VARIANTS[v].items():
                os.environ[k_] = x
            est = BatchEstimator(B, n_states=n)
            for k_ in VARIANTS[v]:
                os.environ.pop(k_)
            est.reset(tv, tq, tP)
            est.run_legodo(d_imu, d_lo, d_mask, q4)
            reps = max(2, int(3e-2 / (K * B * bps / 5e12)))
            ms = min(sum(est.run_legodo(d_imu, d_lo, d_mask, q4, timed=True) for _ in range(reps)) / reps for _ in range(3))
            us = ms / K * 1e3
            rb = est.run_block()
            print("n=%d B=%8d K=%3d %-9s %-24s %8.2f us  %6.0f GB/s  frac %.3f  %s" % (n, B, K, v, est.hot_kernel(), us, bps * B / us / 1e3,
                                                                             bps * B / us / 1e3 / 8000,
                                                                             ("blocked: %d filters per block" % rb) if rb else "step by step"), flush=True)
            est.close()
        del d_imu, d_lo, d_mask, tv, tq, tP

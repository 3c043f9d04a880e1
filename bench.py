#!/usr/bin/env python3
"""bench.py -- EKF predict+update steps/s on MI355X (BASELINE.json metric, configs[1] at N=1).

One step = one IMU predict (RBISIMUProcessStep) + one 3-DoF leg-odometry update (RBISIndexedMeasurement on
idx 3..5) for every filter of the batch = ONE kernel launch; the posterior is written back to HBM after every
step (T = 1 accounting).  Inputs (IMU + leg-odometry streams for warmup+steps) are resident
in HBM before the timed region.  N > 1: one process per GPU (torch.distributed / RCCL), the batch is split by
filter range with NO data-path collective ("weak" scaling: per-GPU batch fixed); the only collective is the
end-of-run summary all-reduce.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "EKF predict+update steps/sec (batched filters), 15-state, 1000 Hz IMU"
HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def bytes_per_step(n):
    """Algorithmic bytes per filter-step, T=1, symmetric-packed P (SURVEY.md 8d / BASELINE.md section 3)."""
    s_x = (n + 4) * 8
    s_p = n * (n + 1) // 2 * 8
    return 2 * (s_x + s_p + 8) + 56 + 48


def cpu_baseline(n, dt_us, target_s, est_cls):
    """Dense op-for-op oracle (kind "port") on this box's host cores, on a bounded sample of the same workload;
    the HIP path is run on the same sample to report the parity of what was just timed."""
    from oracle import po
    from pronto_amd.synth import Workload
    # a GPU box gives one GPU's job a 16-core CPU share whatever nproc says: more OpenMP threads only oversubscribe it
    threads = max(1, min(po.lib().po_max_threads(), int(os.environ.get("PRONTO_CPU_THREADS", "16"))))
    T = 100

    def run(Bs, nthreads=threads):
        w = Workload(Bs, n_states=n, dt_us=dt_us)
        vec, quat, P0 = w.initial_state()
        v21 = np.zeros((21, Bs)); v21[:n] = vec
        P21 = np.zeros((21, 21, Bs)); P21[:n, :n] = P0
        ob = po.OracleBatch(v21, quat, P21)
        imu, lo, mask = w.streams(0, T)
        sec = ob.run_legodo(imu, lo, mask, w.process_noise(), nthreads=nthreads)
        return sec, ob, (w, vec, quat, P0, imu, lo, mask)

    sec1, _, _ = run(256, nthreads=1)  # BASELINE.md section 4 "cpu-dense-1t"
    sec, _, _ = run(4 * threads)  # calibration
    rate = 4 * threads * T / max(sec, 1e-9)
    Bs = int(max(4 * threads, min(262144, rate * target_s / T)))
    Bs = (Bs + threads - 1) // threads * threads
    sec, ob, (w, vec, quat, P0, imu, lo, mask) = run(Bs)
    out = {"value": Bs * T / sec, "unit": "steps/s", "cores": threads, "kind": "port",
           "sample": "%d filters x %d steps of the same synthetic workload, dense 21-state oracle "
                     "(oracle/pronto_oracle.c, gcc -O2 -fopenmp), %.1f s" % (Bs, T, sec),
           "value_1thread": 256 * T / sec1}
    # parity of the HIP path on the very same sample
    import torch
    est = est_cls(Bs, n_states=n, device=torch.cuda.current_device())
    est.set_constants(*po.constants())
    est.reset(vec, quat, P0)
    dev = torch.device("cuda", torch.cuda.current_device())
    est.run_legodo(torch.from_numpy(imu).to(dev), torch.from_numpy(lo).to(dev), torch.from_numpy(mask).to(dev),
                   w.process_noise())
    gv, gq, gP, gll = est.get_head()
    est.close()

    def rel(a, b):
        return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))

    out["parity_max_rel_err"] = max(rel(gv, ob.vec[:n]), rel(gq, ob.quat), rel(gP, ob.cov[:n, :n]), rel(gll, ob.ll))
    try:
        out.update(cpu_structured(n, dt_us, threads, min(3.0, target_s)))
    except Exception as e:  # a reported extra, never a reason to lose the bench line
        out["structured_error"] = repr(e)
    return out


def cpu_structured(n, dt_us, threads, target_s):
    """SURVEY.md 8d asks for the structured CPU figure next to the dense one, so that the GPU/CPU ratio is not inflated
    by the reference's dense waste: the kernels' own per-filter arithmetic (block-sparse Ad, rank-3 downdate, packed P)
    compiled for the host by the TEST harness (tests/host_harness.cpp, g++ -O2), one filter range per host thread."""
    import ctypes as C
    import subprocess
    from concurrent.futures import ThreadPoolExecutor
    from pronto_amd.synth import Workload
    so = os.path.join(ROOT, "tests", "build", "libhost_harness.so")
    src = os.path.join(ROOT, "tests", "host_harness.cpp")
    hdr = os.path.join(ROOT, "pronto_amd", "csrc", "rbis_device.hpp")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-o", so, src])
    hh = C.CDLL(so)
    from oracle import po
    g, tol = po.constants()
    T, Bt = 50, 512
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))

    def prepare(t):
        w = Workload(Bt, b0=t * Bt, n_states=n, dt_us=dt_us)
        vec, quat, P0 = w.initial_state()
        nc = hh.hh_nc(n)
        st = np.zeros((nc, Bt))
        st[:n], st[n:n + 4] = vec, quat
        for i in range(n):
            for j in range(i + 1):
                st[n + 5 + hh.hh_pk(i, j)] = P0[i, j]
        imu, lo, mask = w.streams(0, T)
        return st, np.ascontiguousarray(imu), np.ascontiguousarray(lo), np.ascontiguousarray(mask), \
            np.ascontiguousarray(w.process_noise(), dtype=np.float64)

    def work(job, reps):
        st, imu, lo, mask, q4 = job
        for _ in range(reps):
            for k in range(T):
                hh.hh_step(C.c_int(n), dp(st), C.c_long(Bt), C.c_int(Bt), dp(imu[k]), dp(lo[k]),
                           mask[k].ctypes.data_as(C.POINTER(C.c_uint8)), dp(q4), C.c_double(g), C.c_double(tol), C.c_int(1))

    jobs = [prepare(t) for t in range(threads)]
    t0 = time.perf_counter(); work(jobs[0], 1); one = time.perf_counter() - t0
    reps = max(1, int(target_s / max(one, 1e-6)))
    with ThreadPoolExecutor(threads) as ex:
        t0 = time.perf_counter()
        list(ex.map(lambda j: work(j, reps), jobs))
        sec = time.perf_counter() - t0
    return {"value_structured": threads * Bt * T * reps / sec, "value_structured_1thread": Bt * T / one,
            "structured_sample": "%d threads x %d filters x %d steps, the kernels' block-structured arithmetic compiled "
                                 "for the host (tests/host_harness.cpp), %.1f s" % (threads, Bt, T * reps, sec)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch-per-gpu", type=int, default=65536)
    ap.add_argument("--n-states", type=int, default=15, choices=[15, 21])
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target wall time of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fused", type=int, default=0, metavar="T",
                    help="also time the time-fused replay kernel (T steps per launch, state resident in registers) and "
                         "report it under its own accounting in a 'fused' object; never the headline value")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from pronto_amd.batch import BatchEstimator
    from pronto_amd.shard import allreduce_summary, shard_range
    from pronto_amd.synth import Workload

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py: no GPU visible; the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # PRONTO_BENCH_FORCE_DIST=1 takes the torch.distributed / RCCL path with a single rank too (a rehearsal of the N > 1
    # code on a one-GPU box; launch through torch.distributed.run so that the rendezvous variables exist)
    use_dist = world > 1 or os.environ.get("PRONTO_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    n, K, W = args.n_states, args.steps, args.warmup
    Bper = args.batch_per_gpu
    total = Bper * world
    b0, b1 = shard_range(total, rank, world)
    B = b1 - b0
    dt_us = 1000

    # ---- synthetic streams for this shard, resident in HBM before the timed region ----
    w = Workload(B, b0=b0, n_states=n, dt_us=dt_us)
    vec, quat, P0 = w.initial_state()
    q4 = w.process_noise()
    T = W + K
    d_imu = torch.empty((T, 7, B), dtype=torch.float64, device=dev)
    d_lo = torch.empty((T, 6, B), dtype=torch.float64, device=dev)
    d_mask = torch.empty((T, B), dtype=torch.uint8, device=dev)
    CH = 16
    for s in range(0, T, CH):
        e = min(T, s + CH)
        imu, lo, mask = w.streams(s, e - s)
        d_imu[s:e].copy_(torch.from_numpy(imu))
        d_lo[s:e].copy_(torch.from_numpy(lo))
        d_mask[s:e].copy_(torch.from_numpy(mask))

    est = BatchEstimator(B, n_states=n, device=local_rank)
    est.reset(vec, quat, P0)
    est.sync()

    def barrier():
        if use_dist:
            dist.barrier(device_ids=[local_rank])

    # ---- warmup (untimed) ----
    if W:
        est.run_legodo(d_imu[:W], d_lo[:W], d_mask[:W], q4)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    # ---- timed: exactly K steps = K launches of k_step<15,true> ----
    t0 = time.perf_counter()
    ev_ms = est.run_legodo(d_imu[W:], d_lo[W:], d_mask[W:], q4, timed=True)
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    wall = torch.tensor([t1 - t0, ev_ms * 1e-3], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(wall, op=dist.ReduceOp.MAX)
    wall_s, ev_s = float(wall[0]), float(wall[1])

    summary = allreduce_summary(est.summary(), dist if use_dist else None, dev)

    fused = None
    if args.fused > 0 and n == 15:
        # secondary accounting: bytes_step(T) = 2*(S_x+S_P+8)/T + 56 + 48; the bound is fp64 VALU issue
        Tf = args.fused
        est.reset(vec, quat, P0)
        est.replay_legodo_fused(d_imu[:W], d_lo[:W], d_mask[:W], q4, Tf)
        torch.cuda.synchronize()
        fms = est.replay_legodo_fused(d_imu[W:], d_lo[W:], d_mask[W:], q4, Tf, timed=True)
        torch.cuda.synchronize()
        bst = 2 * ((n + 4) * 8 + n * (n + 1) // 2 * 8 + 8) / Tf + 56 + 48
        fused = {"steps_per_launch": Tf, "value_per_gpu": B * K / (fms * 1e-3), "unit": "steps/s",
                 "bytes_per_filter_step": bst, "achieved_GBps": bst * B * K / (fms * 1e-3) / 1e9,
                 "frac_of_hbm_roofline_under_this_accounting": bst * B * K / (fms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                 "note": "posterior written once per launch; not the plugin path, not the headline metric"}

    if rank == 0:
        bps = bytes_per_step(n)
        value = total * K / wall_s
        launch_s = ev_s / K                       # HIP events on the launch stream around the K launches
        achieved = bps * B / launch_s / 1e9       # algorithmic bytes of ONE launch (this rank's shard) / its duration
        out = {
            "metric": METRIC, "value": value, "unit": "steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": wall_s / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "64k batched 15-state filters, IMU predict + 3-DoF leg-odom update, 1 MI355X"
                       if (n == 15 and Bper == 65536) else
                       "%d batched %d-state filters per GPU, IMU predict + 3-DoF leg-odom update" % (Bper, n),
                       "batch_per_gpu": Bper, "n_states": n, "imu_dt_us": dt_us, "kernel": est.hot_kernel(),
                       "launches_per_step": 1, "bytes_per_filter_step": bps, "parallelism": "filter-range split x%d" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                         "kernel_avg_us": launch_s * 1e6, "algorithmic_bytes_per_launch": bps * B},
            "summary": {"sum_loglik": float(summary[0]), "checksum_abs": float(summary[1]),
                        "max_quat_norm_dev": float(summary[2]), "nonfinite": float(summary[3])},
        }
        tr = os.path.join(ROOT, "profiles", "traffic.json")  # HBM bytes per launch from a separate --pmc pass
        if os.path.exists(tr):
            try:
                rec = json.load(open(tr))
                key = "%s@%d" % (est.hot_kernel(), B)
                if key in rec:
                    out["roofline"]["traffic"] = rec[key]["hbm_bytes_per_launch"]
            except Exception:
                pass
        if fused is not None:
            out["fused"] = fused
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(n, dt_us, args.cpu_seconds, BatchEstimator)
        print(json.dumps(out), flush=True)
    est.close()
    if use_dist:
        dist.barrier(device_ids=[local_rank])
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

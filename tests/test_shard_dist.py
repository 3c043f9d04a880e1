"""Multi-rank path on CPU: world_size-2 gloo (the N>1 bench path uses the same helpers over RCCL)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pronto_amd.shard import allreduce_summary, shard_range
from pronto_amd.synth import Workload


def test_shard_range_partitions_exactly():
    for total in (0, 1, 7, 64, 65536, 262144, 1000003):
        for world in (1, 2, 3, 8):
            rs = [shard_range(total, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
            sizes = [b - a for a, b in rs]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def test_workload_shards_are_slices_of_the_global_workload():
    """Counter-based generator: rank r's shard is bit-identical to the slice of the single-GPU workload."""
    B = 96
    full = Workload(B, n_states=15)
    imu_f, lo_f, mk_f = full.streams(3, 4)
    v_f, q_f, P_f = full.initial_state()
    for r in range(2):
        b0, b1 = shard_range(B, r, 2)
        w = Workload(b1 - b0, b0=b0, n_states=15)
        imu, lo, mk = w.streams(3, 4)
        assert np.array_equal(imu, imu_f[:, :, b0:b1]) and np.array_equal(lo, lo_f[:, :, b0:b1])
        assert np.array_equal(mk, mk_f[:, b0:b1])
        v, q, P = w.initial_state()
        assert np.array_equal(v, v_f[:, b0:b1]) and np.array_equal(P, P_f[:, :, b0:b1])


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b0, b1 = shard_range(1001, rank, world)
    local = [float(b1 - b0), 10.0 * (rank + 1), 1e-15 * (rank + 1), float(rank)]
    out = allreduce_summary(local, dist)
    t = torch.tensor([0.5 + rank])          # the bench's max-over-ranks timing reduction
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    q.put((rank, out.tolist(), t.item()))
    dist.destroy_process_group()


def test_summary_allreduce_world2_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for rank, out, tmax in res:
        assert out == [1001.0, 30.0, 2e-15, 1.0]
        assert tmax == 1.5


def test_bench_spawns_its_own_ranks_dry():
    """`python bench.py --gpus 2` with no launcher starts two child ranks itself (the form the driver uses); the dry
    rehearsal runs the rendezvous + an all-reduce over gloo without a GPU and rank 0 prints the one JSON line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PRONTO_BENCH_REHEARSE="dry")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rank_sum"] == 3.0
    # a failing rank fails the whole run (no GPU here -> every real rank exits non-zero)
    env.pop("PRONTO_BENCH_REHEARSE")
    if not torch.cuda.is_available():
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2"],
                           env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and "rank(s) failed" in r.stderr


def test_reduce_summaries_matches_allreduce_semantics():
    from pronto_amd.shard import reduce_summaries
    assert reduce_summaries([[1.0, 10.0, 1e-15, 0.0], [2.0, 20.0, 3e-15, 1.0]]) == [3.0, 30.0, 3e-15, 1.0]


def test_single_process_is_identity():
    assert allreduce_summary([1.0, 2.0, 3.0, 4.0]).tolist() == [1.0, 2.0, 3.0, 4.0]


def _bench_line(extra_env, *flags):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "PRONTO_BENCH_REHEARSE", "PRONTO_BENCH_FORCE_DIST"):
        env.pop(k, None)
    env.update(extra_env)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "6", "--warmup", "2", "--min-timed-ms", "0",
                        "--batch-per-gpu", "4096", "--no-cpu-baseline", "--no-cache-busting", *flags],
                       capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_two_ranks_on_one_card_equal_one_rank_of_twice_the_batch():
    """The N > 1 code of bench.py on the one-GPU test box: two self-spawned ranks share GPU 0 and rendezvous over gloo
    (PRONTO_BENCH_REHEARSE=1); their reduced summary must equal the summary of ONE rank that runs both shards as one
    batch (the generator is counter-based: shard r is a slice of the global workload)."""
    two = _bench_line({"PRONTO_BENCH_REHEARSE": "1"}, "--gpus", "2")
    one = _bench_line({}, "--batch-per-gpu", "8192")
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1
    a, b = two["summary"], one["summary"]
    assert abs(a["sum_loglik"] - b["sum_loglik"]) <= 1e-9 * abs(b["sum_loglik"])
    assert abs(a["checksum_abs"] - b["checksum_abs"]) <= 1e-12 * abs(b["checksum_abs"])
    assert a["nonfinite"] == 0 and b["nonfinite"] == 0


@pytest.mark.gpu
def test_bench_rccl_path_with_a_single_rank():
    """PRONTO_BENCH_FORCE_DIST=1: init_process_group("nccl") = RCCL, barrier and the summary all-reduce on the device,
    with world size 1 -- the transport the multi-GPU run uses, as far as one card can exercise it."""
    out = _bench_line({"PRONTO_BENCH_FORCE_DIST": "1", "MASTER_PORT": "29547"})
    assert out["n_gpus"] == 1 and out["summary"]["nonfinite"] == 0 and out["value"] > 0


@pytest.mark.gpu
def test_bench_four_ranks_on_one_card_rehearsal_is_quick_and_equals_one_rank():
    """More ranks than the two-rank rehearsal: four self-spawned ranks x 8192 filters on GPU 0 (the GPU box allows at most six
    processes on its card at once, this test runner included, so eight cannot be rehearsed here) against one rank with all
    32 768 filters; every rank makes its streams on the device (per_rank_ms reports the spread), and start-to-JSON stays far
    below the two minutes eight ranks are allowed."""
    import time
    t0 = time.time()
    four = _bench_line({"PRONTO_BENCH_REHEARSE": "1"}, "--gpus", "4", "--batch-per-gpu", "8192")
    took = time.time() - t0
    one = _bench_line({}, "--batch-per-gpu", "32768")
    assert four["n_gpus"] == 4 and took < 120, took
    pr = four["per_rank_ms"]
    assert pr["streams"].startswith("device") and 0 < pr["kernel_time_min"] <= pr["kernel_time_max"]
    assert pr["stream_generation_max"] < 30e3
    a, b = four["summary"], one["summary"]
    assert abs(a["sum_loglik"] - b["sum_loglik"]) <= 1e-9 * abs(b["sum_loglik"])
    assert abs(a["checksum_abs"] - b["checksum_abs"]) <= 1e-12 * abs(b["checksum_abs"])
    assert a["nonfinite"] == 0
    # the host generator gives the same workload (to rounding of the device's sin / log)
    host = _bench_line({}, "--batch-per-gpu", "32768", "--host-streams")
    assert abs(host["summary"]["checksum_abs"] - b["checksum_abs"]) <= 1e-9 * abs(b["checksum_abs"])


def test_bench_names_its_workload_truthfully_at_every_gpu_count():
    """config.workload: BASELINE.json's wording only where the run IS that config; N > 1 lines say how many
    filters each GPU holds, how many there are in all and which scaling mode ran."""
    import bench
    assert bench.workload_name(15, 65536, 65536, 1, "weak") == "64k batched 15-state filters, IMU predict + 3-DoF leg-odom update, 1 MI355X"
    w8 = bench.workload_name(15, 65536, 8 * 65536, 8, "weak")
    assert "1 MI355X" not in w8 and "x 8 MI355X" in w8 and "512k filters" in w8 and "weak" in w8
    s8 = bench.workload_name(15, 32768, 262144, 8, "strong")
    assert "256k" in s8 and "32k per GPU" in s8 and "strong" in s8 and "configs[3]" in s8
    assert "configs[3]" not in bench.workload_name(15, 131072, 262144, 2, "strong")
    assert "21-state" in bench.workload_name(21, 65536, 65536, 1, "weak")


@pytest.mark.gpu
def test_bench_strong_scaling_mode_splits_one_job():
    """--scaling strong: --total-batch filters split over the ranks (BASELINE config 4's shape), two ranks rehearsed on one card
    against one rank with the whole job; the line says which mode ran."""
    two = _bench_line({"PRONTO_BENCH_REHEARSE": "1"}, "--gpus", "2", "--scaling", "strong", "--total-batch", "8192")
    one = _bench_line({}, "--scaling", "strong", "--total-batch", "8192")
    assert two["scaling"] == "strong" and two["config"]["batch_per_gpu"] == 4096 and two["config"]["total_batch"] == 8192
    assert "strong" in two["config"]["workload"] and one["config"]["batch_per_gpu"] == 8192
    a, b = two["summary"], one["summary"]
    assert abs(a["sum_loglik"] - b["sum_loglik"]) <= 1e-9 * abs(b["sum_loglik"])
    assert abs(a["checksum_abs"] - b["checksum_abs"]) <= 1e-12 * abs(b["checksum_abs"])

"""Batch split across ranks: filters are independent, so the data path has NO collective.
Each rank owns a contiguous range of global filter ids; the only exchange is the end-of-run summary
all-reduce (sum of log-likelihoods / checksum / non-finite count, max of |q|^2-1)."""


def shard_range(total, rank, world):
    """Contiguous [b0, b1) of `total` filters for `rank`; remainders go to the lowest ranks."""
    if not (0 <= rank < world) or total < 0:
        raise ValueError("bad rank/world/total")
    base, rem = divmod(total, world)
    b0 = rank * base + min(rank, rem)
    return b0, b0 + base + (1 if rank < rem else 0)


def allreduce_summary(summary4, dist=None, device=None):
    """summary4 = per-shard [sum_ll, checksum, max_qdev, nonfinite] -> job-wide, via torch.distributed
    (RCCL on GPUs, gloo in the CPU tests).  Without an initialised process group this is the identity."""
    import torch
    t = torch.as_tensor(summary4, dtype=torch.float64)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return t.clone()
    if device is not None:
        t = t.to(device)
    sums = t.clone()
    mx = t.clone()
    dist.all_reduce(sums, op=dist.ReduceOp.SUM)
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    out = sums.clone()
    out[2] = mx[2]
    return out.cpu()


def reduce_summaries(shard_summaries):
    """What the all-reduce computes, on a plain list of per-shard summaries (the C++ multi-device driver and the
    one-GPU rehearsal of config 4 use this order: sums in rank order, max of the quaternion-norm deviation)."""
    out = [0.0, 0.0, 0.0, 0.0]
    for s in shard_summaries:
        out[0] += float(s[0])
        out[1] += float(s[1])
        out[2] = max(out[2], float(s[2]))
        out[3] += float(s[3])
    return out

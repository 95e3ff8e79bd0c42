"""Multi-GPU plumbing: streams are independent, so ranks shard them with no data-path collective.

The only cross-rank traffic is the timing barrier and the MAX-over-ranks reduction of bench.py
(RCCL on the GPU box via backend "nccl"; "gloo" in the CPU tests)."""


def shard_range(total_streams, world, rank):
    """Contiguous, balanced [lo, hi) of the streams rank `rank` owns (first `total % world` ranks get one more)."""
    if world < 1 or not 0 <= rank < world or total_streams < 0:
        raise ValueError("bad shard arguments")
    q, r = divmod(total_streams, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def max_over_ranks(value, dist=None, device=None):
    """MAX of a python float over all ranks (identity without a process group)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, dist=None, device=None):
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())

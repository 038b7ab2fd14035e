"""Batch sharding across the GPUs of one node.  Every PBS of a batch is independent, keys are
replicated on every GPU by seeded keygen, so rank r simply owns a contiguous index range.  For PBS batches (bench.py)
that is all: no data-path collective, no RCCL traffic; torch.distributed only carries the timing barrier / MAX
reduction.  The encrypted inverse (executor.py) uses the same ranges per level and adds ONE all-gather per split level -
the path's only exchange step."""
from __future__ import annotations


def shard_range(total: int, rank: int, world: int):
    """contiguous [start, stop) of `total` items owned by `rank` (sizes differ by at most one)"""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def reduce_max(value: float, device=None) -> float:
    """MAX over ranks of a host scalar (elapsed seconds); identity when torch.distributed is not initialised"""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()

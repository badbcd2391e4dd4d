"""Database sharding over GPUs: one process per GPU, contiguous shards of the db file
order, no collective on the data path; ONE gather of the per-shard score arrays to rank 0
at the end of a search (RCCL over xGMI on GPUs; the same code runs over gloo on CPU
tensors in the tests).  The reference is single-GPU (cudaSaTabsearch.cu:790 "TODO allow
multiple GPUs"); this is the multi-GPU mode of the new build.

Every (query, db entry) pair is independent and the random streams are keyed by the
entry's ordinal in the whole database, so the gathered result is identical for any
number of shards.
"""
import numpy as np


def shard_bounds(total, world):
    """Contiguous, near-equal shards: [begin_0, ..., begin_world] over 0..total.  The cost of
    scoring an entry is dominated by the query size, so equal counts balance the work."""
    return [(total * r) // world for r in range(world + 1)]


def shard_range(total, world, rank):
    b = shard_bounds(total, world)
    return b[rank], b[rank + 1]


def gather_to_rank0(local, total, world, rank, dist=None, group=None):
    """Gather 1-D per-shard tensors (shard sizes from shard_bounds) into one tensor of
    length `total` on rank 0; other ranks get None.  Shards are padded to the largest
    shard so that a single fixed-size gather does the exchange."""
    import torch
    if world == 1:
        return local
    bounds = shard_bounds(total, world)
    width = max(bounds[r + 1] - bounds[r] for r in range(world))
    padded = local
    if local.shape[0] != width:
        padded = torch.zeros(width, dtype=local.dtype, device=local.device)
        padded[:local.shape[0]] = local
    gather_list = [torch.empty_like(padded) for _ in range(world)] if rank == 0 else None
    dist.gather(padded, gather_list, dst=0, group=group)
    if rank != 0:
        return None
    return torch.cat([gather_list[r][:bounds[r + 1] - bounds[r]] for r in range(world)])

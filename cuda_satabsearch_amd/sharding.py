"""Database sharding over GPUs: one process per GPU, contiguous cost-balanced shards of the db file
order, no collective on the data path; ONE gather of the per-shard score arrays to rank 0
at the end of a search (RCCL over xGMI on GPUs; the same code runs over gloo on CPU
tensors in the tests).  The reference is single-GPU (cudaSaTabsearch.cu:790 "TODO allow
multiple GPUs"); this is the multi-GPU mode of the new build.

Every (query, db entry) pair is independent and the random streams are keyed by the
entry's ordinal in the whole database, so the gathered result is identical for any
number of shards.
"""
import numpy as np


def entry_cost(orders):
    """Relative cost of scoring each entry (1.0 at 32 SSEs): the measured table of
    csrc/host/sat_shard.c (kernel time per scoring by entry order)."""
    from . import _native
    host = _native.host_lib()
    return np.array([host.sat_entry_cost(int(o)) for o in np.asarray(orders).ravel()])


def shard_bounds(total, world, orders=None):
    """Contiguous shards [begin_0, ..., begin_world] over 0..total.  With `orders` (the number of SSEs
    of every entry, file order) the cuts balance the shards' COST - real databases are size sorted and a
    96-SSE entry costs four 32-SSE ones - through the same C routine the command line and the
    multi-GPU C API use (sat_shard_cuts); without, near-equal counts (entries of one size)."""
    if orders is None:
        return [(total * r) // world for r in range(world + 1)]
    from . import _native
    orders = np.ascontiguousarray(orders, dtype=np.int32)
    assert orders.shape[0] == total
    begin = np.zeros(world + 1, np.int32)
    if _native.host_lib().sat_shard_cuts(int(total), orders.ctypes.data, int(world), begin.ctypes.data) != 0:
        raise ValueError("sat_shard_cuts failed")
    return [int(b) for b in begin]


def shard_range(total, world, rank, orders=None):
    b = shard_bounds(total, world, orders)
    return b[rank], b[rank + 1]


def gather_to_rank0(local, total, world, rank, dist=None, group=None, orders=None):
    """Gather 1-D per-shard tensors (shard sizes from shard_bounds) into one tensor of
    length `total` on rank 0; other ranks get None.  Shards are padded to the largest
    shard so that a single fixed-size gather does the exchange."""
    import torch
    if world == 1:
        return local
    bounds = shard_bounds(total, world, orders)
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

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


class ShardGather:
    """The one exchange of a sharded search, set up once and then repeated: every rank's score row goes to
    rank 0, which holds the whole database's rows in file order.  All buffers are allocated here - a
    padded send row per slot and, on rank 0, `world` receive rows per slot - and there are two slots,
    so that the gather of search k (asynchronous: on RCCL it runs on the communicator's own stream)
    overlaps search k + 1, which writes the context's score buffer again.  The same code runs over
    gloo on CPU tensors (the tests, and rehearsals on a box with fewer GPUs than ranks)."""

    SLOTS = 2

    def __init__(self, bounds, rank, dist, device, dtype=None, group=None):
        import torch
        self.bounds = [int(b) for b in bounds]
        self.world = len(self.bounds) - 1
        self.rank = rank
        self.dist = dist
        self.group = group
        self.n_local = self.bounds[rank + 1] - self.bounds[rank]
        self.width = max(self.bounds[r + 1] - self.bounds[r] for r in range(self.world))
        dtype = dtype or torch.int32
        self.send = [torch.zeros(self.width, dtype=dtype, device=device) for _ in range(self.SLOTS)]
        self.recv = [[torch.empty(self.width, dtype=dtype, device=device) for _ in range(self.world)]
                     if rank == 0 else None for _ in range(self.SLOTS)]
        self.work = [None] * self.SLOTS
        self.k = 0

    def start(self, local):
        """Queue the gather of `local` (this rank's n_local scores; a device tensor for RCCL, anything
        copyable into the send row otherwise).  Returns the slot; the row is read when the copy into the
        slot's send buffer runs, so the caller may overwrite `local` with work queued after this call."""
        b = self.k % self.SLOTS
        self.k += 1
        if self.work[b] is not None:
            self.work[b].wait()                   # the gather that last used this slot (two searches ago)
        # (asynchronous only device to device: a non-blocking copy into pageable host memory would not be ordered
        # with the gather that reads it)
        self.send[b][:self.n_local].copy_(local[:self.n_local], non_blocking=local.device.type == self.send[b].device.type)
        self.work[b] = self.dist.gather(self.send[b], self.recv[b], dst=0, group=self.group, async_op=True)
        return b

    def wait(self, slot=None):
        for b in (range(self.SLOTS) if slot is None else [slot]):
            if self.work[b] is not None:
                self.work[b].wait()
                self.work[b] = None

    def rows(self, slot):
        """Rank 0: the gathered rows of `slot` in database order (a new tensor of bounds[-1] scores)."""
        import torch
        self.wait(slot)
        if self.rank != 0:
            return None
        return torch.cat([self.recv[slot][r][:self.bounds[r + 1] - self.bounds[r]] for r in range(self.world)])

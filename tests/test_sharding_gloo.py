"""N > 1 path on CPU: two processes over gloo, each holding one contiguous shard of the
database; per-shard scores are gathered to rank 0 with the same
cuda_satabsearch_amd.sharding code bench.py runs over RCCL.  There is no CPU search in
the product, so the oracle (Philox streams, keyed by db ordinal) stands in for the
kernel here: what is under test is sharding, ordinals, padding and the gather."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import cuda_satabsearch_amd as sat
    import oracle_lib
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = sat.sharding.shard_range(total, world, rank)
    shard = sat.synth.make_db(hi - lo, 6, 20, first_index=lo, total=total)      # generated per rank
    q = sat.synth.planted_query(sat.synth.make_db(total, 6, 20), total - 3)
    scores, _, _ = oracle_lib.search(shard, *q, True, False, 32, db_ordinal=np.arange(lo, hi))
    gathered = sat.sharding.gather_to_rank0(torch.from_numpy(scores), total, world, rank, dist)
    if rank == 0:
        np.save(out_path, gathered.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [101, 64])      # unequal (padded) and equal shards
def test_two_rank_gather_equals_single_process(tmp_path, total):
    import torch.multiprocessing as mp
    import cuda_satabsearch_amd as sat
    import oracle_lib
    out = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(2, _free_port(), total, out), nprocs=2, join=True)
    gathered = np.load(out)
    db = sat.synth.make_db(total, 6, 20)
    q = sat.synth.planted_query(db, total - 3)
    whole, _, _ = oracle_lib.search(db, *q, True, False, 32)
    assert gathered.shape == (total,)
    assert np.array_equal(gathered, whole)
    assert gathered.argmax() == total - 3


def test_shard_bounds_cover_everything():
    import cuda_satabsearch_amd as sat
    for total in (1, 7, 8, 1000, 1_000_000):
        for world in (1, 2, 3, 8):
            b = sat.sharding.shard_bounds(total, world)
            assert b[0] == 0 and b[-1] == total and all(b[i] <= b[i + 1] for i in range(world))
            assert max(b[i + 1] - b[i] for i in range(world)) - min(b[i + 1] - b[i] for i in range(world)) <= 1

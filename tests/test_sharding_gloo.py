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
    # size-sorted database: the cuts are placed by cumulative COST, so the shards differ in length
    orders = sat.synth._orders(total, 6, 20, sat.synth.DB_SEED, True)
    lo, hi = sat.sharding.shard_range(total, world, rank, orders)
    shard = sat.synth.make_db(hi - lo, 6, 20, first_index=lo, total=total)      # generated per rank
    q = sat.synth.planted_query(sat.synth.make_db(total, 6, 20), total - 3)
    scores, _, _ = oracle_lib.search(shard, *q, True, False, 32, db_ordinal=np.arange(lo, hi))
    gathered = sat.sharding.gather_to_rank0(torch.from_numpy(scores), total, world, rank, dist, orders=orders)
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



def test_cost_balanced_cuts_on_size_sorted_databases():
    """real databases are size sorted, so the shards are cut by cumulative cost
    (the measured per-order table of csrc/host/sat_shard.c), not by count: on the sorted C3 (orders
    8..32) and C5 (8..111, 1 % above 96) order distributions the most expensive shard costs at most
    1.1 x the cheapest for 2, 4 and 8 GPUs - where equal counts are off by up to 2.5 x."""
    import cuda_satabsearch_amd as sat
    n = 100_000
    c3 = sat.synth._orders(n, 8, 32, sat.synth.DB_SEED, True)
    c5 = sat.synth.orders_c5(n)
    for name, orders in (("C3", c3), ("C5", c5)):
        cost = sat.sharding.entry_cost(orders)
        for world in (2, 4, 8):
            b = sat.sharding.shard_bounds(n, world, orders)
            assert b[0] == 0 and b[-1] == n and all(b[i] < b[i + 1] for i in range(world))
            shard_cost = [cost[b[g]:b[g + 1]].sum() for g in range(world)]
            assert max(shard_cost) / min(shard_cost) <= 1.1, (name, world, shard_cost)
            eq = sat.sharding.shard_bounds(n, world)
            eq_cost = [cost[eq[g]:eq[g + 1]].sum() for g in range(world)]
            if name == "C5" and world == 8:
                assert max(eq_cost) / min(eq_cost) > 3.0          # what equal counts would have given
    # the cost model itself: 1.0 at 32 SSEs, monotone, the measured end points
    assert sat.sharding.entry_cost([32])[0] == 1.0
    c = sat.sharding.entry_cost(np.arange(1, 112))
    assert (np.diff(c) >= 0).all() and 0.4 < c[0] < 0.6 and 4.0 < c[-1] < 4.8      # 46.9 / 88.9 and 394.0 / 88.9 ns (round 3)


def test_cuts_degenerate_cases():
    import cuda_satabsearch_amd as sat
    assert sat.sharding.shard_bounds(3, 3, np.array([5, 100, 5])) == [0, 1, 2, 3]      # every shard non-empty
    assert sat.sharding.shard_bounds(8, 2, np.array([111] + [4] * 7)) in ([0, 1, 8], [0, 2, 8])
    assert sat.sharding.shard_bounds(5, 1, np.arange(1, 6)) == [0, 5]


def _gather_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import cuda_satabsearch_amd as sat
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bounds = [0, 37, 101]                                   # unequal shards: the shorter row is padded
    g = sat.sharding.ShardGather(bounds, rank, dist, "cpu")
    n = bounds[rank + 1] - bounds[rank]
    local = torch.zeros(n, dtype=torch.int32)
    got = []
    for k in range(5):                                      # five "searches": the two slots are reused
        local[:] = torch.arange(bounds[rank], bounds[rank + 1], dtype=torch.int32) * (k + 1)
        slot = g.start(local)
        local[:] = -1                                       # the next search overwrites the score row at once
        if k >= 3:
            rows = g.rows(slot)
            if rank == 0:
                got.append(rows.numpy().copy())
    g.wait()
    if rank == 0:
        np.save(out_path, np.stack(got))
    dist.barrier()
    dist.destroy_process_group()


def test_preallocated_double_buffered_gather(tmp_path):
    """sharding.ShardGather (bench.py's exchange): buffers allocated once, two slots reused step after step,
    the send row copied out before the caller's score row is overwritten, rows back in database order."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "rows.npy")
    mp.spawn(_gather_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    assert got.shape == (2, 101)
    assert np.array_equal(got[0], np.arange(101) * 4) and np.array_equal(got[1], np.arange(101) * 5)


def test_bench_starts_its_own_ranks_and_relays_their_exit_code():
    """`python bench.py --gpus 2` with no torch.distributed environment must start the ranks itself (the parent
    never touches the GPU).  Here there is no GPU: the ranks say so and the launcher hands their failure on
    - no hang, no silent success, no result line."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    import torch
    if torch.cuda.is_available():
        pytest.skip("covered by the gpu rehearsal test on a GPU box")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode != 0
    assert "needs a HIP device" in p.stderr
    assert not any(l.lstrip().startswith("{") for l in p.stdout.splitlines())

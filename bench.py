#!/usr/bin/env python3
"""bench.py - db-structure scorings/sec of the SA tableau search on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling weak|strong]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[3] / north_star target shape): one 32-SSE synthetic query against a
synthetic database of 32-SSE structures, r = 128 restarts x 100 SA steps per (query, entry) pair,
LTYPE = T, LORDER = T, LSOLN = F.  The database is sharded contiguously, one shard per GPU / process;
a STEP is one full search of the query over every shard followed by the one gather of the per-shard
score arrays to rank 0 (RCCL over xGMI when N > 1; the gather of step k overlaps search k + 1).  Inputs
are resident in HBM before the timed region.

  --scaling weak    (default) 125 000 entries per GPU: N = 8 is the 1 M-entry configuration
  --scaling strong  ONE database of --total entries (default 1 000 000 = BASELINE configs[3]) cut into N
                    shards by cuda_satabsearch_amd.sharding.shard_bounds

Launching.  `python bench.py --gpus N` with N > 1 and no torch.distributed environment starts the N ranks
itself: this process never touches the GPU, runs `python -m torch.distributed.run --nproc-per-node N
bench.py ...` as a child, relays its stdout (the one JSON line) and exits with its return code.  Under
torch.distributed.run (WORLD_SIZE set) it is one of the ranks.  `--single-process` instead times the
product's own multi-GPU entry points (sat_multi_*: one host thread, one context per GPU, ncclCommInitAll +
ncclGather into device 0) in this process.

Prints ONE JSON line (rank 0).  `value` = (entries of the whole database x K) / max-over-ranks time.
Extra objects: `roofline` (HBM-nominal, see DESIGN.md section 4: the path is VALU/LDS bound, the HBM
fraction is reported because the contract asks for it) and, at N = 1, `cpu_baseline` (the oracle's
restatement of the reference's host path timed on this box's CPU on a bounded sample; the reference's
own objects, when oracle/_ref travelled, under `cpu_baseline_reference`).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PER_GPU_ENTRIES = 125_000
STRONG_TOTAL = 1_000_000
ORDER = 32
MAXSTART = 128
MAXITER = 100
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8 TB/s spec


def algorithmic_bytes_per_scoring(n2):
    # 1 B code + 4 B distance per lower-triangle cell, + order + score
    return 5 * n2 * (n2 + 1) // 2 + 4 + 4


# ------------------------------------------------------------------------------------------------
# CPU baseline legs (the only users of oracle/ in this file)
def cpu_baseline(db, q, sample_seconds=15.0):
    """The oracle's C restatement of the reference's host path (oracle/sa_oracle.c, kind "port") on a
    bounded prefix of the same database: one thread, one sequential drand48 stream - the literal `-c`
    semantics.  Returns (baseline, n) - n = entries of the sample."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    qt, qd, qtypes = q
    t0 = time.time()
    oracle_lib.search(db, qt, qd, qtypes, True, False, MAXSTART, mode=oracle_lib.RNG_DRAND48, entries=np.arange(64))
    per_entry = (time.time() - t0) / 64
    n = int(max(256, min(len(db), sample_seconds / max(per_entry, 1e-6))))
    sample = f"first {n} entries of the rank-0 shard, same query, r={MAXSTART}, single sequential drand48 stream"
    t0 = time.time()
    oracle_lib.search(db, qt, qd, qtypes, True, False, MAXSTART, mode=oracle_lib.RNG_DRAND48, entries=np.arange(n))
    dt = time.time() - t0
    return {"value": n / dt, "unit": "db-structure scorings/sec", "cores": 1, "kind": "port",
            "sample": sample + " (oracle/sa_oracle.c, gcc -O3)"}, n


def cpu_baseline_reference(db, q, n):
    """The reference's own host objects (oracle/_ref/ref_oracle: its unmodified sources compiled in the
    build container; the directory travels to the GPU box with the snapshot but is not in a fresh clone)
    on the same sample.  None when they are not there."""
    import numpy as np
    import cuda_satabsearch_amd as sat
    qt, qd, qtypes = q
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "ref_oracle")
    if not os.path.exists(ref_bin):
        return None
    with tempfile.TemporaryDirectory() as tmp:
        sub = db.subset(np.arange(n))
        sat.synth.write_ascii(sub, os.path.join(tmp, "db.ascii"))
        qset = sat.StructSet.from_dense([len(qtypes)], [qt], [qd], ["SYNQ32"])
        sat.synth.write_ascii(qset, os.path.join(tmp, "q.body"))
        with open(os.path.join(tmp, "q.input"), "w") as f:
            f.write("db.ascii\nT T F\n")
            f.write(open(os.path.join(tmp, "q.body")).read())
        with open(os.path.join(tmp, "q.input")) as fin:
            p = subprocess.run([ref_bin, "-c", "-r", str(MAXSTART)], stdin=fin, cwd=tmp,
                               stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
        ms = [float(l.split()[3]) for l in p.stderr.splitlines() if l.startswith("host execution time")]
        if p.returncode == 0 and ms:
            return {"value": n / (sum(ms) / 1e3), "unit": "db-structure scorings/sec", "cores": 1, "kind": "reference",
                    "sample": f"first {n} entries of the rank-0 shard, same query, r={MAXSTART} (reference sources "
                              "compiled into oracle/_ref, g++ -O3)"}
    return None


def cpu_baseline_all_cores(db, q, seconds=8.0):
    """Throughput of the oracle port on every host core of this box: one thread per contiguous
    chunk of a bounded prefix, each with its own drand48 stream (ctypes releases the GIL).  Not
    byte-comparable to `-c` (neither is any parallel run); the single-thread figure above is."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    qt, qd, qtypes = q
    # one GPU's share of the host on the bench boxes is 16 cores, whatever cpu_count() says
    cores = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    t0 = time.time()
    oracle_lib.search(db, qt, qd, qtypes, True, False, MAXSTART, mode=oracle_lib.RNG_DRAND48, entries=np.arange(32))
    per_entry = (time.time() - t0) / 32
    per_core = int(max(32, min(len(db) // cores, seconds / max(per_entry, 1e-6))))
    chunks = [np.arange(c * per_core, (c + 1) * per_core) for c in range(cores)]

    def work(c):
        oracle_lib.search(db, qt, qd, qtypes, True, False, MAXSTART, mode=oracle_lib.RNG_DRAND48, entries=chunks[c],
                          lcg=oracle_lib.lib().sa_oracle_srand48(1234 + c))
    t0 = time.time()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(work, range(cores)))
    dt = time.time() - t0
    return {"value": cores * per_core / dt, "unit": "db-structure scorings/sec", "cores": cores, "kind": "port",
            "sample": f"{cores} threads x {per_core} entries of the rank-0 shard, same query, r={MAXSTART}, one drand48 stream per thread"}


def oracle_sample_ok(scores, total, q, k=16):
    """Spot check of the TIMED search inside the bench: k random entries of the whole database, generated
    again from their index, scored by the oracle on the kernel's own Philox streams (keyed by the entry's
    ordinal) - must equal what the GPUs returned, bit for bit."""
    import numpy as np
    import cuda_satabsearch_amd as sat
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    qt, qd, qtypes = q
    idx = np.sort(np.random.default_rng(20251005).choice(total, size=min(k, total), replace=False))
    for g in idx:
        one = sat.synth.make_db(1, ORDER, ORDER, first_index=int(g), total=total)
        want, _, _ = oracle_lib.search(one, qt, qd, qtypes, True, False, MAXSTART, db_ordinal=np.array([int(g)]))
        if int(want[0]) != int(scores[int(g)]):
            return False, [int(i) for i in idx]
    return True, [int(i) for i in idx]


# ------------------------------------------------------------------------------------------------
def committed_counters(n_local, kernel_info):
    """HBM traffic and VALU instructions per launch from the rocprofv3 PMC passes of this same command
    (profiles/bench_traffic.json, written by scripts/summarize_prof.py): counters need profiler passes of
    their own and cannot be read inside this run.  The file is stamped with the kernel instantiation and the
    hash of the kernel sources it was measured on: a figure from another build is dropped, not reported."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "bench_traffic.json")))
    except (OSError, ValueError):
        return None, "no profiles/bench_traffic.json"
    from cuda_satabsearch_amd import build
    if t.get("entries_per_launch") != n_local:
        return None, "committed counters are for %s entries per launch" % t.get("entries_per_launch")
    if t.get("kernel_source_sha256") != build.kernel_source_hash():
        return None, "committed counters were measured on other kernel sources (profiles/bench_traffic.json is stale)"
    name = (t.get("kernel") or "").replace(" ", "")
    if not name or name not in kernel_info.replace(" ", ""):
        return None, "committed counters belong to kernel %r" % t.get("kernel")
    return t, None


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args, argv):
    """`python bench.py --gpus N` without a torch.distributed environment: start the N ranks as children.  This
    process has not imported torch and never touches the GPU - no exec of a process that has."""
    if not args.all_ranks_on_device0:
        # (counting devices does not initialise the GPU; everything else about it is left to the ranks)
        try:
            import torch
            have = torch.cuda.device_count()
        except Exception:
            have = None
        if have is not None and 0 < have < args.gpus:
            print("bench.py: --gpus %d but only %d HIP device(s) visible (rehearse the N > 1 flow on fewer GPUs with "
                  "--backend gloo --all-ranks-on-device0)" % (args.gpus, have), file=sys.stderr)
            sys.exit(2)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    got_line = False
    for line in p.stdout:
        # stdout carries the ONE result line; whatever else the ranks' libraries print there (gloo announces its
        # connections on stdout) goes to stderr
        is_result = line.lstrip().startswith("{")
        (sys.stdout if is_result else sys.stderr).write(line)
        (sys.stdout if is_result else sys.stderr).flush()
        got_line = got_line or is_result
    rc = p.wait()
    if rc == 0 and not got_line:
        print("bench.py: the ranks exited 0 without a result line", file=sys.stderr)
        rc = 1
    sys.exit(rc)


def parse_args(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)      # 3.2 s of searches at N = 1
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--entries", type=int, default=PER_GPU_ENTRIES, help="weak scaling: db entries per GPU")
    ap.add_argument("--total", type=int, default=STRONG_TOTAL, help="strong scaling: entries of the whole database")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-upload-probe", action="store_true", help="skip the one-off upload + first search measurement "
                    "(profiler runs: its piece-wise launches would be averaged into the kernel's counters)")
    ap.add_argument("--no-regimes", action="store_true", help="skip the all-hit / planted-query side rates")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse "
                    "the N > 1 flow on a box with fewer GPUs than ranks)")
    ap.add_argument("--all-ranks-on-device0", action="store_true", help="rehearsal only: every rank uses GPU 0")
    ap.add_argument("--single-process", action="store_true", help="time the product's sat_multi_* entry points (one host "
                    "thread, ncclCommInitAll + ncclGather) instead of one process per GPU")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------
def regimes(sat, np, device):
    """Side rates of the same kernel class outside the all-miss regime the headline runs in (a random query
    matches little): a database in which EVERY entry is a near copy of the query's source (all hits, dense
    maps - the worst case of the work compaction) and a planted query (a member of the database with a quarter
    of its SSEs removed) against random entries.  20 000 entries each, r = 128, 3 timed searches."""
    n = 20_000
    out = {}
    base = sat.synth.make_db(1, ORDER, ORDER, seed=5)
    t, d = base.dense(0)
    hits = sat.StructSet.from_dense([ORDER] * n, [t] * n, [d] * n, ["h%06d" % i for i in range(n)])
    rnd = sat.synth.make_db(n, ORDER, ORDER, seed=6)
    q_hit = sat.synth.planted_query(base, 0, keep=1.0, jitter=0.5)
    q_planted = sat.synth.planted_query(rnd, n // 2)
    with sat.Searcher(device) as s:
        for name, db, q in (("all_hit", hits, q_hit), ("planted_query", rnd, q_planted)):
            s.upload(db)
            s.set_query(*q, 0)
            s.search_timed(True, False, MAXSTART, 1)
            tot, _ = s.search_timed(True, False, MAXSTART, 3)
            out[name + "_scorings_per_sec"] = n / (tot / 3) * 1e3
            out[name + "_query_sses"] = int(q[0].shape[0])
    out["note"] = "20 000 32-SSE entries each, r=128; all_hit: every entry a copy of the query's source; " \
                  "planted_query: a db member minus 25 % of its SSEs against random entries"
    return out


def run_single_process(args):
    """The product's own multi-GPU path: sat_multi_* from ONE host thread (contexts on args.gpus devices,
    cost-balanced shards, one gather into device 0, rows to the host in database order)."""
    import numpy as np
    import cuda_satabsearch_amd as sat
    ndev = args.gpus
    if sat.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device (there is no CPU path to measure)")
    devices = [0] * ndev if args.all_ranks_on_device0 else list(range(ndev))
    total = args.total if args.scaling == "strong" else args.entries * ndev
    db = sat.synth.make_db(total, ORDER, ORDER)
    q = sat.synth.make_query(ORDER)
    with sat.MultiSearcher(ndev, devices) as m:
        m.upload(db)
        m.set_queries([q], 0)
        for _ in range(args.warmup):
            m.search(True, False, MAXSTART)
        t0 = time.perf_counter()
        wall = []
        for _ in range(args.steps):
            scores, _, ms = m.search(True, False, MAXSTART)
            wall.append(ms)
        elapsed = time.perf_counter() - t0
        ok, sample = oracle_sample_ok(scores[0], total, q)
        out = result_header(args, ndev, total, elapsed)
        out["config"]["parallelism"] = "db-shard x%d, one process (sat_multi_*, gather: %s)" % (ndev, m.gather_kind)
        out["ranks_seen"] = ndev
        out["shards"] = [int(b) for b in m.shards()]
        out["search_wall_ms_avg"] = float(np.mean(wall))
        out["oracle_sample_ok"] = bool(ok)
        out["oracle_sample"] = sample
        out["note"] = "value includes the device-0 -> host copy of the rows and their re-ordering (sat_multi_search returns host rows)"
    print(json.dumps(out), flush=True)
    if not ok:
        raise SystemExit("timed search differs from the oracle on the sampled entries")


def result_header(args, world, total, elapsed):
    metric = "db-structure scorings/sec (query×db pairs/sec) at r=128; 1/2/4/8 MI355X"
    try:
        metric = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except (OSError, ValueError, KeyError):
        pass
    value = total * args.steps / elapsed
    return {
        "metric": metric,
        "value": value, "unit": "db-structure scorings/sec", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "i32",
        "dtype_note": "integer pair scores; f32 distance comparisons and u8 tableau codes feed them",
        "data": "synthetic (seeded generator, cuda_satabsearch_amd/synth.py)",
        "config": {"workload": "32-SSE synthetic query x %d-entry synthetic db (32 SSEs per entry; %s), "
                               "r=128 restarts x 100 SA steps, LTYPE=T LORDER=T LSOLN=F, contiguous db shards, "
                               "one gather of int32 scores per step" %
                               (total, "%d per GPU" % (total // world) if args.scaling == "weak" else "ONE database cut into %d shards" % world),
                   "query_sses": ORDER, "db_entries": total, "restarts": MAXSTART,
                   "parallelism": "db-shard x%d" % world},
        "sa_steps_per_sec": value * MAXSTART * MAXITER,
    }


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    in_dist_env = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if args.single_process:
        return run_single_process(args)
    if not in_dist_env and args.gpus > 1:
        return launch_ranks(args, argv)

    import numpy as np
    import torch
    import cuda_satabsearch_amd as sat

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU path to measure)")
    if args.all_ranks_on_device0:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    # ---- inputs: this rank's contiguous shard of the synthetic database + the query
    if args.scaling == "strong":
        total = args.total
        bounds = sat.sharding.shard_bounds(total, world)          # one order: equal cost = equal counts
    else:
        total = args.entries * world
        bounds = [args.entries * r for r in range(world + 1)]
    lo, hi = bounds[rank], bounds[rank + 1]
    n_local = hi - lo
    db = sat.synth.make_db(n_local, ORDER, ORDER, first_index=lo, total=total)
    q = sat.synth.make_query(ORDER)
    qt, qd, qtypes = q
    searcher = sat.Searcher(local_rank)
    t_up = time.perf_counter()
    searcher.upload(db, db_ordinal=np.arange(lo, hi))
    upload_ms = (time.perf_counter() - t_up) * 1e3        # host -> HBM of the packed shard (synchronous copies)
    searcher.set_query(qt, qd, qtypes, 0)
    # a single query over a freshly read shard, upload included: the copy and the first search overlapped
    # (sat_db_upload_search: each piece of the shard is searched while the next one is copied), on a
    # context of its own; wall time of the call, which returns when the scores are complete on the GPU
    overlapped_ms = None
    if world == 1 and not args.no_upload_probe:
        with sat.Searcher(local_rank) as one_shot:
            one_shot.set_query(qt, qd, qtypes, 0)
            one_shot.upload_search(db, True, False, MAXSTART, db_ordinal=np.arange(n_local))     # first call of the process
            times = []
            for _ in range(5):
                t1 = time.perf_counter()
                one_shot.upload_search(db, True, False, MAXSTART, db_ordinal=np.arange(n_local))
                times.append((time.perf_counter() - t1) * 1e3)
            overlapped_ms = float(np.median(times))
            first_scores, _ = one_shot.results()
    # launch on torch's current stream: the gather and the timing events follow the kernel
    searcher.use_stream(torch.cuda.current_stream().cuda_stream)
    # the device score buffer is asked for AFTER a search has been queued (satabsearch.h: pointer
    # lifetime); it then stays where it is until the next upload / query change
    searcher.search_async(True, False, MAXSTART)
    scores_dev = searcher.device_scores_tensor()
    on_gpu = args.backend == "nccl"
    gather = None
    if world > 1:
        # every buffer of the exchange allocated once; two slots, so that gather k overlaps search k + 1
        gather = sat.sharding.ShardGather(bounds, rank, dist, scores_dev.device if on_gpu else "cpu")

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    slot = None
    for _ in range(args.warmup):
        searcher.search_async(True, False, MAXSTART)
        if gather:
            slot = gather.start(scores_dev)
    if gather:
        gather.wait()
    fence()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[2 * k].record()                       # same stream the SA kernel is launched on
        searcher.search_async(True, False, MAXSTART)
        ev[2 * k + 1].record()
        if gather:
            slot = gather.start(scores_dev)
    if gather:
        gather.wait()
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms = [ev[2 * k].elapsed_time(ev[2 * k + 1]) for k in range(args.steps)]
    kavg_ms = float(np.mean(kernel_ms))

    ranks_seen, gathered_len, kernel_ms_by_rank, gather_ms = 1, n_local, [kavg_ms], None
    whole = None
    if world > 1:
        dev = "cuda" if on_gpu else "cpu"
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        mine = torch.tensor([kavg_ms], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        kernel_ms_by_rank = [float(x.item()) for x in every]
        ranks_seen = dist.get_world_size()
        # the gather on its own (no search to hide behind): barrier, G gathers, device sync
        fence()
        tg = time.perf_counter()
        for _ in range(10):
            s_ = gather.start(scores_dev)
            gather.wait(s_)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - tg) / 10 * 1e3
        rows = gather.rows(s_)
        if rank == 0:
            whole = rows.cpu().numpy()
            gathered_len = int(whole.shape[0])
            # the gathered array is the whole database in file order: shard 0 must be rank 0's own scores
            assert gathered_len == total
            own, _ = searcher.results()                    # rank 0's shard, copied by the library itself
            assert np.array_equal(whole[:n_local], own)
    elif rank == 0:
        whole, _ = searcher.results()
    if rank == 0 and overlapped_ms is not None:
        assert np.array_equal(first_scores, whole), "overlapped upload + search differs from upload, then search"

    rc = 0
    if rank == 0:
        out = result_header(args, world, total, elapsed)
        ok, sample = oracle_sample_ok(whole, total, q)
        abytes = algorithmic_bytes_per_scoring(ORDER) * n_local
        achieved = abytes / (kavg_ms * 1e-3) / 1e9
        kernel_info = searcher.last_launch_info()
        t, why_not = committed_counters(n_local, kernel_info)
        traffic = t["hbm_bytes_per_launch"] if t else None
        out.update({
            "ranks_seen": ranks_seen, "gathered_scores": gathered_len,
            "kernel_ms_by_rank": {"max": max(kernel_ms_by_rank), "min": min(kernel_ms_by_rank), "all": kernel_ms_by_rank},
            "gather_ms_alone": gather_ms,
            "oracle_sample_ok": bool(ok), "oracle_sample": sample,
            # the boundary takes host buffers: one-off H->D of the shard, and the rate a single
            # query would see with that copy included (never `value`)
            "h2d_upload_ms": upload_ms,
            "scorings_per_sec_incl_upload_single_query_sequential": total / (elapsed / args.steps + upload_ms * 1e-3),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": ("profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on "
                                            "these kernel sources, scripts/profile_bench.sh)" % t.get("source")) if t else why_not,
                         # the instantiation the library reports for the launches it just made
                         "kernel": kernel_info, "kernel_ms_avg": kavg_ms,
                         "algorithmic_bytes_per_launch": abytes,
                         "note": "nominal bound only: 2648 B per scoring against 12 800 dependent SA steps; "
                                 "the kernel is VALU/LDS-issue bound (DESIGN.md section 4)"},
        })
        if t and t.get("valu_wave_instr_per_launch"):
            # what actually binds: VALU issue.  1024 SIMDs x one wave64 instr per 2 clk at 2.4 GHz
            rate = t["valu_wave_instr_per_launch"] / (kavg_ms * 1e-3)
            out["binding_resource"] = {
                "bound": "valu_issue", "achieved": rate, "peak": 1024 * 2.4e9 / 2, "unit": "wave-instr/s",
                "frac": rate / (1024 * 2.4e9 / 2), "lds_busy_frac": t.get("lds_busy_frac"),
                "note": "VALU wave-instructions per launch from the committed rocprofv3 PMC pass / live kernel time; "
                        "the mix is ~55 % half-rate opcodes (profiles/r01_gfx950_valu_opcode_cost.txt), so ~0.65 "
                        "of this nominal peak is the practical ceiling"}
        if overlapped_ms is not None:
            out["upload_and_search_overlapped_ms"] = overlapped_ms
            out["scorings_per_sec_incl_upload_single_query"] = total / (overlapped_ms * 1e-3)
        if world == 1 and not args.no_regimes:
            out["other_regimes"] = regimes(sat, np, local_rank)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], n_cpu = cpu_baseline(db, q)
            ref = cpu_baseline_reference(db, q, n_cpu)
            if ref:
                out["cpu_baseline_reference"] = ref
            out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(db, q)
        print(json.dumps(out), flush=True)
        if not ok:
            print("bench.py: the timed search differs from the oracle on the sampled entries", file=sys.stderr)
            rc = 1

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    searcher.close()
    sys.exit(rc)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py - db-structure scorings/sec of the SA tableau search on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[3] / north_star target shape): one 32-SSE synthetic
query against a synthetic database of 125 000 x N structures of 32 SSEs (N = 8 is the
1M-entry configuration), r = 128 restarts x 100 SA steps per (query, entry) pair,
LTYPE = T, LORDER = T, LSOLN = F.  The database is sharded contiguously, one shard per
GPU / process; a STEP is one full search of the query over every shard followed by the
one gather of the per-shard score arrays to rank 0 (RCCL over xGMI when N > 1).  Inputs
are resident in HBM before the timed region.

Prints ONE JSON line (rank 0).  `value` = (N x 125 000 x K) / max-over-ranks time.
Extra objects: `roofline` (HBM-nominal, see DESIGN.md section 4: the path is VALU/LDS
bound, the HBM fraction is reported because the contract asks for it) and, at N = 1,
`cpu_baseline` (the reference's host path timed on this box's CPU on a bounded sample).
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PER_GPU_ENTRIES = 125_000
ORDER = 32
MAXSTART = 128
MAXITER = 100
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8 TB/s spec


def algorithmic_bytes_per_scoring(n2):
    # 1 B code + 4 B distance per lower-triangle cell, + order + score
    return 5 * n2 * (n2 + 1) // 2 + 4 + 4


def cpu_baseline(db, q, sample_seconds=15.0):
    """Time the reference host path on this machine's CPU on a bounded prefix of the same
    database.  Prefers the reference's own sources compiled into oracle/_ref (kind
    "reference"); otherwise the C restatement in oracle/ (kind "port").  One thread, one
    sequential drand48 stream: the literal `-c` semantics."""
    import cuda_satabsearch_amd as sat
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    qt, qd, qtypes = q
    # size the sample from a short calibration run of the port
    t0 = time.time()
    oracle_lib.search(db, qt, qd, qtypes, True, False, MAXSTART, mode=oracle_lib.RNG_DRAND48,
                      entries=np.arange(64))
    per_entry = (time.time() - t0) / 64
    n = int(max(256, min(len(db), sample_seconds / max(per_entry, 1e-6))))
    sample = f"first {n} entries of the rank-0 shard, same query, r={MAXSTART}, single sequential drand48 stream"
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "ref_oracle")
    if os.path.exists(ref_bin):
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
                return {"value": n / (sum(ms) / 1e3), "unit": "db-structure scorings/sec", "cores": 1,
                        "kind": "reference", "sample": sample + " (reference sources compiled into oracle/_ref, g++ -O3)"}
    t0 = time.time()
    oracle_lib.search(db, qt, qd, qtypes, True, False, MAXSTART, mode=oracle_lib.RNG_DRAND48, entries=np.arange(n))
    dt = time.time() - t0
    return {"value": n / dt, "unit": "db-structure scorings/sec", "cores": 1, "kind": "port",
            "sample": sample + " (oracle/sa_oracle.c, gcc -O3)"}


def cpu_baseline_all_cores(db, q, seconds=8.0):
    """Throughput of the oracle port on every host core of this box: one thread per contiguous
    chunk of a bounded prefix, each with its own drand48 stream (ctypes releases the GIL).  Not
    byte-comparable to `-c` (neither is any parallel run); the single-thread figure above is."""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)      # 3.4 s of searches at N = 1
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--entries", type=int, default=PER_GPU_ENTRIES, help="db entries per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-upload-probe", action="store_true", help="skip the one-off upload + first search measurement "
                    "(profiler runs: its piece-wise launches would be averaged into the kernel's counters)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse "
                    "the N > 1 flow on a box with fewer GPUs than ranks)")
    ap.add_argument("--all-ranks-on-device0", action="store_true", help="rehearsal only: every rank uses GPU 0")
    args = ap.parse_args()

    import torch
    import cuda_satabsearch_amd as sat

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
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
    n_local = args.entries
    total = n_local * world
    db = sat.synth.make_db(n_local, ORDER, ORDER, first_index=rank * n_local, total=total)
    qt, qd, qtypes = sat.synth.make_query(ORDER)
    searcher = sat.Searcher(local_rank)
    t_up = time.perf_counter()
    searcher.upload(db, db_ordinal=np.arange(rank * n_local, (rank + 1) * n_local))
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
    # launch on torch's current stream: the RCCL gather and the timing events follow the kernel
    searcher.use_stream(torch.cuda.current_stream().cuda_stream)
    # the device score buffer is asked for AFTER a search has been queued (satabsearch.h: pointer
    # lifetime); it then stays where it is until the next upload / query change
    searcher.search_async(True, False, MAXSTART)
    scores_dev = searcher.device_scores_tensor()

    def gather():
        if args.backend == "nccl":
            return sat.sharding.gather_to_rank0(scores_dev, total, world, rank, dist)
        return sat.sharding.gather_to_rank0(scores_dev.cpu(), total, world, rank, dist)     # rehearsal path

    def step():
        searcher.search_async(True, False, MAXSTART)
        if world > 1:
            gather()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[2 * k].record()                       # same stream the SA kernel is launched on
        searcher.search_async(True, False, MAXSTART)
        ev[2 * k + 1].record()
        if world > 1:
            gathered = gather()
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms = [ev[2 * k].elapsed_time(ev[2 * k + 1]) for k in range(args.steps)]

    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        if rank == 0:
            # the gathered array is the whole database in file order: shard 0 must be rank 0's own scores
            assert gathered.shape[0] == total
            own, _ = searcher.results()                    # rank 0's shard, copied by the library itself
            assert np.array_equal(gathered[:n_local].cpu().numpy(), own)
    if rank == 0 and overlapped_ms is not None:
        own, _ = searcher.results()
        assert np.array_equal(first_scores, own), "overlapped upload + search differs from upload, then search"

    if rank == 0:
        scorings = total * args.steps
        value = scorings / elapsed
        kavg_ms = float(np.mean(kernel_ms))
        abytes = algorithmic_bytes_per_scoring(ORDER) * n_local
        achieved = abytes / (kavg_ms * 1e-3) / 1e9
        # HBM traffic per launch from the rocprofv3 PMC passes of this same command (profiles/)
        traffic = None
        traffic_source = None
        issue = None
        try:
            t = json.load(open(os.path.join(ROOT, "profiles", "bench_traffic.json")))
            if t.get("entries_per_launch") == n_local:
                traffic = t["hbm_bytes_per_launch"]
                # PMC counters need rocprofv3 passes of their own (the guide's recipe): they cannot be read
                # inside this run, so the figure is the committed one of the same command and kernel
                traffic_source = "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, scripts/profile_bench.sh)" % t.get("source")
                if t.get("valu_wave_instr_per_launch"):
                    # what actually binds: VALU issue.  1024 SIMDs x one wave64 instr per 2 clk at 2.4 GHz
                    rate = t["valu_wave_instr_per_launch"] / (kavg_ms * 1e-3)
                    issue = {"bound": "valu_issue", "achieved": rate, "peak": 1024 * 2.4e9 / 2, "unit": "wave-instr/s",
                             "frac": rate / (1024 * 2.4e9 / 2), "lds_busy_frac": t.get("lds_busy_frac"),
                             "note": "VALU wave-instructions per launch from the committed rocprofv3 PMC pass / live kernel time; "
                                     "the mix is ~55 % half-rate opcodes (profiles/r01_gfx950_valu_opcode_cost.txt), so ~0.65 "
                                     "of this nominal peak is the practical ceiling"}
        except (OSError, ValueError, KeyError):
            pass
        metric = "db-structure scorings/sec (query\u00d7db pairs/sec) at r=128; 1/2/4/8 MI355X"
        try:
            metric = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
        except (OSError, ValueError, KeyError):
            pass
        out = {
            "metric": metric,
            "value": value, "unit": "db-structure scorings/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "i32",
            "dtype_note": "integer pair scores; f32 distance comparisons and u8 tableau codes feed them",
            "data": "synthetic (seeded generator, cuda_satabsearch_amd/synth.py)",
            "config": {"workload": "32-SSE synthetic query x %d-entry synthetic db (32 SSEs per entry; %d per GPU), "
                                   "r=128 restarts x 100 SA steps, LTYPE=T LORDER=T LSOLN=F, contiguous db shards, "
                                   "one gather of int32 scores per step" % (total, n_local),
                       "query_sses": ORDER, "db_entries": total, "restarts": MAXSTART,
                       "parallelism": "db-shard x%d" % world},
            "sa_steps_per_sec": value * MAXSTART * MAXITER,
            # the boundary takes host buffers: one-off H->D of the shard, and the rate a single
            # query would see with that copy included (never `value`)
            "h2d_upload_ms": upload_ms,
            "scorings_per_sec_incl_upload_single_query_sequential": total / (elapsed / args.steps + upload_ms * 1e-3),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": traffic_source,
                         # the instantiation the library reports for the launches it just made
                         "kernel": searcher.last_launch_info(), "kernel_ms_avg": kavg_ms,
                         "algorithmic_bytes_per_launch": abytes,
                         "note": "nominal bound only: 2648 B per scoring against 12 800 dependent SA steps; "
                                 "the kernel is VALU/LDS-issue bound (DESIGN.md section 4)"},
        }
        if overlapped_ms is not None:
            out["upload_and_search_overlapped_ms"] = overlapped_ms
            out["scorings_per_sec_incl_upload_single_query"] = total / (overlapped_ms * 1e-3)
        if issue:
            out["binding_resource"] = issue
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(db, (qt, qd, qtypes))
            out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(db, (qt, qd, qtypes))
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    searcher.close()


if __name__ == "__main__":
    main()

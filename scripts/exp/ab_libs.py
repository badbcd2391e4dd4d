"""A/B runs of several builds of the device library on ONE workload, in one GPU-box call:
    python scripts/exp/ab_libs.py WORKLOAD ROUNDS lib1.so lib2.so ...
WORKLOAD: bench (32-SSE query x 125 000 32-SSE entries), lorderf (same, LORDER = F, 40 000 entries), q101 (101-SSE
query x 30 000 entries of 8..96 SSEs), n96 (32-SSE query x 20 000 96-SSE entries), n64, mixed (8..32 sorted), c4 (BASELINE
configs[4]: 101-SSE query x C5 orders, LSOLN).  The parent builds the inputs once (npz in /tmp); every (round, library)
is a fresh process (the library is chosen at import, SAT_DEVICE_LIB) that times 5 searches with HIP events; the rounds
interleave the libraries so that clock drift hits all alike.  Prints one line per run and the per-library medians."""
import os, subprocess, sys, json, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

CHILD = r'''
import os, sys, json
sys.path.insert(0, %r)
import numpy as np
import cuda_satabsearch_amd as sat
z = np.load(sys.argv[1], allow_pickle=False)
db = sat.StructSet(z["orders"], ["e%%d" %% i for i in range(len(z["orders"]))], z["cell_off"], z["tab"], z["dist"])
lorder, lsoln, r = bool(int(sys.argv[2])), bool(int(sys.argv[3])), int(sys.argv[4])
with sat.Searcher(0) as s:
    s.upload(db); s.set_query(z["qt"], z["qd"], z["qtypes"], 0)
    s.search_timed(lorder, lsoln, r, 1)
    tot, _ = s.search_timed(lorder, lsoln, r, 5)
    sc, _, _ = s.search(lorder, lsoln, r)
    print(json.dumps({"ms": tot / 5, "checksum": int(sc.astype(np.int64).sum()), "kernels": s.last_launch_info()}))
''' % ROOT


def workload(name):
    import numpy as np
    import cuda_satabsearch_amd as sat
    from cuda_satabsearch_amd import workloads
    lorder, lsoln, r = True, False, 128
    if name == "bench":
        db, q = sat.synth.make_db(125_000, 32, 32), sat.synth.make_query(32)
    elif name == "lorderf":
        db, q, lorder = sat.synth.make_db(40_000, 32, 32), sat.synth.make_query(32), False
    elif name == "q101":
        db, q = sat.synth.make_db(30_000, 8, 96), workloads.config4_query()[1:]
    elif name == "q101s":
        db, q = sat.synth.make_db(60_000, 8, 32), workloads.config4_query()[1:]
    elif name == "n96":
        db, q = sat.synth.make_db(20_000, 96, 96), sat.synth.make_query(32)
    elif name == "n40":
        db, q = sat.synth.make_db(60_000, 40, 40), sat.synth.make_query(32)
    elif name == "n48":
        db, q = sat.synth.make_db(50_000, 48, 48), sat.synth.make_query(32)
    elif name == "n64":
        db, q = sat.synth.make_db(40_000, 64, 64), sat.synth.make_query(32)
    elif name == "mixed":
        db, q = workloads.mixed_db(100_000), sat.synth.make_query(32)
    elif name == "c4":
        db, q, lsoln = workloads.config4_db(100_000), workloads.config4_query()[1:], True
    elif name == "c2q101":        # the 101-SSE query of configs[2] at its restart count, on a tenth of its database
        db, q, r = workloads.config2_db(10_000), workloads.config4_query()[1:], 4096
    elif name == "c2q19":
        db, q, r = workloads.config2_db(10_000), workloads.config2_queries()[0][1:], 4096
    elif name == "c2q8":
        db, q, r = workloads.config2_db(10_000), workloads.config2_queries()[1][1:], 4096
    elif name.startswith("qn"):       # qnNN: a synthetic NN-SSE query x 60 000 sorted entries of 4..40 SSEs (the query list's database shape)
        db, q = sat.synth.make_db(60_000, 4, 40, sort=True), sat.synth.make_query(int(name[2:]))
    elif name.startswith("real"):     # realNN: an NN-SSE query x 200 000 sorted entries of 3..32 SSEs with the reference database's skew
        import numpy as np
        rng = np.random.default_rng(11)
        # the example database: 71 % of the entries up to 16 SSEs, 24 % 17..32, median 11 (tests/golden/inputs)
        orders = np.sort(np.clip(np.round(rng.gamma(2.6, 5.2, size=200_000)).astype(np.int32), 3, 32))
        db, q = sat.synth.make_db(200_000, orders=orders), sat.synth.make_query(int(name[4:]))
    elif name == "q16":
        db, q = sat.synth.make_db(100_000, 8, 32), sat.synth.make_query(12)
    else:
        raise SystemExit("unknown workload " + name)
    return db, q, lorder, lsoln, r


def main():
    import numpy as np
    name, rounds, libs = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
    db, q, lorder, lsoln, r = workload(name)
    path = os.path.join(tempfile.gettempdir(), "ab_%s.npz" % name)
    np.savez(path, orders=db.orders, cell_off=db.cell_off, tab=db.tab, dist=db.dist, qt=q[0], qd=q[1], qtypes=q[2])
    child = os.path.join(tempfile.gettempdir(), "ab_child.py")
    open(child, "w").write(CHILD)
    res = {l: [] for l in libs}
    sums = {}
    for k in range(rounds):
        for l in (libs if k % 2 == 0 else libs[::-1]):
            # "lib.so#NAME=value,NAME=value": the same build under launch-heuristic overrides (satabsearch_debug.h)
            libpath, _, extra = l.partition("#")
            env = dict(os.environ, SAT_DEVICE_LIB=os.path.join(ROOT, libpath))
            env.update(dict(kv.split("=", 1) for kv in extra.split(",") if kv))
            p = subprocess.run([sys.executable, child, path, str(int(lorder)), str(int(lsoln)), str(r)], capture_output=True, text=True, env=env)
            if p.returncode != 0:
                print(l, "FAILED", p.stderr[-500:], flush=True)
                continue
            o = json.loads(p.stdout.strip().splitlines()[-1])
            res[l].append(o["ms"])
            sums.setdefault(o["checksum"], []).append(l)
            print(f"{name} round {k} {l}: {o['ms']:.3f} ms  {len(db) / o['ms'] * 1e3 / 1e6:.3f} M/s  [{o['kernels'][:90]}]", flush=True)
    print("checksums:", {k: sorted(set(v)) for k, v in sums.items()} if len(sums) > 1 else "all equal")
    for l in libs:
        if res[l]:
            m = float(np.median(res[l]))
            print(f"MEDIAN {name} {l}: {m:.3f} ms = {len(db) / m * 1e3 / 1e6:.3f} M scorings/s")


if __name__ == "__main__":
    main()

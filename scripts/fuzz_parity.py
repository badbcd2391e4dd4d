"""Randomized GPU-vs-oracle parity soak (30 s of it with a fixed seed run inside the suite, tests/test_gpu_parity.py): many random databases,
queries, option combinations and restart counts; stops at the first mismatch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import cuda_satabsearch_amd as sat
import oracle_lib

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
t0 = time.time()
cases = 0
last_note = t0
if True:
    while time.time() - t0 < budget:
        n = int(rng.integers(3, 60))
        lo = int(rng.integers(1, 40)); hi = int(rng.integers(lo, min(111, lo + int(rng.integers(1, 80))) + 1))
        db = sat.synth.make_db(n, lo, hi, seed=int(rng.integers(1, 1 << 30)), sort=bool(rng.integers(0, 2)))
        # sprinkle '?' codes and exact-4.0 differences
        tab = db.tab.copy(); dist = db.dist.copy()
        off = np.nonzero(tab > 3)[0]
        if off.size:
            k = rng.choice(off, size=min(off.size, 5), replace=False); tab[k] = 0x44
        db = sat.StructSet(db.orders, db.names, db.cell_off, tab, np.round(dist * 2) / 2 if rng.random() < 0.3 else dist)
        src = int(rng.integers(0, n))
        keep = float(rng.choice([0.5, 0.8, 1.0]))
        if db.orders[src] < 2:
            continue
        q = sat.synth.planted_query(db, src, keep=keep, jitter=float(rng.choice([0.0, 0.5, 3.9])), seed=int(rng.integers(1, 1 << 30)))
        lorder, lsoln = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        r = int(rng.choice([1, 7, 64, 65, 128, 200, 300]))
        qord = int(rng.integers(0, 5))
        for k, v in (("SAT_EXP_LPC", rng.choice(["", "0", "1", "2"])), ("SAT_EXP_COMPACT", rng.choice(["", "0", "1"])),
                     ("SAT_EXP_EPW", rng.choice(["", "", "2", "3", "4"]))):
            if v: os.environ[k] = str(v)
            else: os.environ.pop(k, None)
        # the launch-heuristic overrides are read when a context is created: one context per case
        with sat.Searcher(0, seed=77) as s:
            s.upload(db)
            s.set_query(*q, qord)
            sc, mp, _ = s.search(lorder, lsoln, r)
            if rng.random() < 0.3:                       # the best-k rows of the same search
                k = int(rng.integers(1, n + 3))
                hits = s.topk_hits(k)
                order = np.lexsort((np.arange(n), -sc.astype(np.int64)))[:min(k, n)]
                if not (np.array_equal(hits[0]["entry"], order) and np.array_equal(hits[0]["score"], sc[order])):
                    print("TOPK MISMATCH", dict(n=n, k=k))
                    sys.exit(1)
        osc, omp, _ = oracle_lib.search(db, *q, lorder, lsoln, r, seed=77, query_ordinal=qord)
        ok = np.array_equal(sc, osc) and (not lsoln or np.array_equal(mp, omp))
        cases += 1
        if time.time() - last_note > 45:                 # a line a minute: a silent GPU run is taken to be hung
            print(f"  {cases} cases ok after {time.time()-t0:.0f}s", flush=True)
            last_note = time.time()
        if not ok:
            print("MISMATCH", dict(n=n, lo=lo, hi=hi, src=src, keep=keep, lorder=lorder, lsoln=lsoln, r=r, qord=qord,
                                   env={k: os.environ.get(k) for k in ("SAT_EXP_LPC", "SAT_EXP_COMPACT", "SAT_EXP_EPW")}))
            print(np.nonzero(sc != osc)[0][:10], sc[sc != osc][:10], osc[sc != osc][:10])
            sys.exit(1)
print(f"fuzz ok: {cases} random cases in {time.time()-t0:.0f}s")

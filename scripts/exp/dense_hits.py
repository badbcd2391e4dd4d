"""Worst case for the work compaction: a database in which EVERY entry is a near copy of the query's
source structure (all hits, dense maps) against the usual all-misses scan."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import cuda_satabsearch_amd as sat
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
base = sat.synth.make_db(1, 32, 32, seed=5)
t, d = base.dense(0)
db = sat.StructSet.from_dense([32] * n, [t] * n, [d] * n, ["h%06d" % i for i in range(n)])
rnd = sat.synth.make_db(n, 32, 32, seed=6)
q = sat.synth.planted_query(base, 0, keep=1.0, jitter=0.5)
with sat.Searcher(0) as s:
    for name, x in (("all hits", db), ("all misses", rnd)):
        s.upload(x); s.set_query(*q, 0)
        s.search_timed(True, False, 128, 1)
        tot, _ = s.search_timed(True, False, 128, 3)
        sc, _, _ = s.search(True, False, 128)
        print(f"{name}: {tot/3:.3f} ms -> {n/(tot/3)*1e3:,.0f} scorings/s, mean score {sc.mean():.1f}")

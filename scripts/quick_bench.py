"""Quick single-GPU throughput probe: python scripts/quick_bench.py [entries] [n1] [n2lo] [n2hi] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cuda_satabsearch_amd as sat
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
n1 = int(sys.argv[2]) if len(sys.argv) > 2 else 32
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 32
hi = int(sys.argv[4]) if len(sys.argv) > 4 else lo
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
db = sat.synth.make_db(n, lo, hi)
q = sat.synth.make_query(n1)
with sat.Searcher(0) as s:
    s.upload(db); s.set_query(*q, 0)
    s.search_timed(True, False, 128, 1)
    tot, _ = s.search_timed(True, False, 128, reps)
    sc, _, _ = s.search(True, False, 128)
    print(f"n={n} n1={n1} n2=[{lo},{hi}] r=128: {tot/reps:.3f} ms/search -> {n/(tot/reps)*1e3:,.0f} scorings/s  checksum {int(sc.astype(np.int64).sum())} env={ {k:v for k,v in os.environ.items() if k.startswith('SAT_')} }")

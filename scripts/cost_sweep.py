"""Per-entry cost by entry order: time fixed-order databases (python scripts/cost_sweep.py) for the
shard-balancing cost table of cuda_satabsearch_amd/csrc/host/sat_shard.c."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cuda_satabsearch_amd as sat
from cuda_satabsearch_amd import workloads as w

queries = {"q32": w.config3_query(), "q8": sat.synth.make_query(8), "q19": w.config2_queries()[0][1:], "q101": w.config4_query()[1:]}
orders = [4, 8, 12, 16, 20, 24, 28, 32, 40, 48, 56, 64, 72, 80, 88, 96, 104, 111]
print("order " + " ".join(f"{k:>9s}" for k in queries))
with sat.Searcher(0) as s:
    for n2 in orders:
        n = max(4000, min(60000, int(60000 * 32 * 32 / max(n2 * n2, 256))))
        db = sat.synth.make_db(n, n2)
        s.upload(db)
        row = []
        for name, q in queries.items():
            s.set_query(*q, 0)
            s.search_timed(True, False, 128, 1)
            tot, _ = s.search_timed(True, False, 128, 3)
            row.append(tot / 3 / n * 1e6)          # ns per scoring
        print(f"{n2:5d} " + " ".join(f"{v:9.1f}" for v in row), flush=True)

"""Print the launch plan (instantiation, grid, block, LDS) of the named workloads: python scripts/exp/launch_info.py n96 n64 ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import cuda_satabsearch_amd as sat
from cuda_satabsearch_amd import workloads as w

for name in sys.argv[1:]:
    lorder, lsoln, r = True, False, 128
    if name == "n96": db, queries = sat.synth.make_db(2000, 96), [w.config3_query()]
    elif name == "n64": db, queries = sat.synth.make_db(4000, 64), [w.config3_query()]
    elif name == "n48": db, queries = sat.synth.make_db(4000, 48), [w.config3_query()]
    elif name == "c4": db, queries, lsoln = w.config4_db(20000), [w.config4_query()[1:]], True
    elif name == "q101": db, queries = sat.synth.make_db(20000, 8, 96, sort=True), [w.config4_query()[1:]]
    elif name == "c2": db, queries, r = w.config2_db(20000), [(t, d, ty) for _, t, d, ty in w.config2_queries()], 4096
    else: raise SystemExit(name)
    with sat.Searcher(0) as s:
        s.upload(db); s.set_queries(queries, 0)
        s.search(lorder, lsoln, r)
        print(name, "->")
        for part in s.last_launch_info().split("; "):
            print("   ", part)

#!/usr/bin/env python3
"""scripts/exp/wpl_split_probe.py - is a query class worth splitting by words per lane?  Times the 8- and
13-SSE queries of BASELINE configs[2] (one size class, two round shapes: the batch runs the WPL = 0
instantiation) together and one by one (each alone runs the instantiation specialised for its shape)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import cuda_satabsearch_amd as sat
from cuda_satabsearch_amd import workloads as w

db = w.config2_db()
qs = [(t, d, ty) for _, t, d, ty in w.config2_queries()]
small = [q for q in qs if q[0].shape[0] <= 16]
print("orders of the small-class queries:", [q[0].shape[0] for q in small])
r = int(sys.argv[1]) if len(sys.argv) > 1 else 512
with sat.Searcher(0) as s:
    s.upload(db)
    def t(queries):
        s.set_queries(queries, 0)
        s.search_timed(True, False, r, 1)
        tot, _ = s.search_timed(True, False, r, 3)
        return tot / 3, s.last_launch_info()
    both, info = t(small)
    print("batch of %d: %.2f ms  %s" % (len(small), both, info.split(" grid")[0]))
    tot = 0.0
    for q in small:
        ms, info = t([q])
        tot += ms
        print("alone (%d SSEs): %.2f ms  %s" % (q[0].shape[0], ms, info.split(" grid")[0]))
    print("sum of singles %.2f ms vs batch %.2f ms: %+.1f %%" % (tot, both, (both / tot - 1) * 100))

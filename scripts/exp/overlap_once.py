#!/usr/bin/env python3
"""scripts/exp/overlap_once.py [pieces] - two overlapped upload + search calls on the bench shard (for a profiler / SAT_EXP_UPLOAD_TIMING=1)"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
if len(sys.argv) > 1:
    os.environ["SAT_EXP_UPLOAD_PIECES"] = sys.argv[1]
import cuda_satabsearch_amd as sat

db = sat.synth.make_db(125000, 32, 32)
q = sat.synth.make_query(32)
with sat.Searcher(0) as s:
    s.set_query(*q, 0)
    for k in range(3):
        t = time.perf_counter()
        s.upload_search(db, True, False, 128)
        print("call %d: %.2f ms" % (k, (time.perf_counter() - t) * 1e3), file=sys.stderr, flush=True)

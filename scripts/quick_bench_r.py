"""Throughput vs restart count (workgroup size / occupancy probe)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cuda_satabsearch_amd as sat
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
db = sat.synth.make_db(n, 32)
q = sat.synth.make_query(32)
with sat.Searcher(0) as s:
    s.upload(db); s.set_query(*q, 0)
    for r in [int(x) for x in sys.argv[2:]] or [64, 128, 256, 512]:
        s.search_timed(True, False, r, 1)
        tot, _ = s.search_timed(True, False, r, 2)
        ms = tot / 2
        print(f"r={r}: {ms:.3f} ms -> {n/ms*1e3:,.0f} scorings/s, {n*r*100/ms*1e3/1e9:.2f} G steps/s")

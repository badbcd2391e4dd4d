"""LSOLN / LORDER / restart-count variants of the headline shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cuda_satabsearch_amd as sat
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
db = sat.synth.make_db(n, 32)
q = sat.synth.make_query(32)
with sat.Searcher(0) as s:
    s.upload(db); s.set_query(*q, 0)
    for lorder, lsoln, r in [(True, False, 128), (True, True, 128), (False, False, 128), (False, True, 128), (True, False, 4096)]:
        if r > 1000:
            s.upload(sat.synth.make_db(6144, 32))      # 4 full rounds of 1536 resident workgroups
        s.search_timed(lorder, lsoln, r, 1)
        tot, _ = s.search_timed(lorder, lsoln, r, 2)
        ms = tot / 2
        print(f"lorder={lorder} lsoln={lsoln} r={r}: {ms:.3f} ms -> {s.n_entries/ms*1e3:,.0f} scorings/s, {s.n_entries*r*100/ms*1e3/1e9:.2f} G steps/s")

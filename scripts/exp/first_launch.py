import sys, time; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cuda_satabsearch_amd as sat
db = sat.synth.make_db(50000, 8, 32); q = sat.synth.make_query(32)
t0=time.time()
with sat.Searcher(0) as s:
    t1=time.time(); s.upload(db); s.set_query(*q, 0); t2=time.time()
    for i in range(3):
        t=time.time(); sc,_,ms = s.search(True, False, 128); print(f"search {i}: wall {1e3*(time.time()-t):.2f} ms, kernel window {ms:.2f} ms")
print(f"ctx create {1e3*(t1-t0):.1f} ms, upload {1e3*(t2-t1):.1f} ms")

"""Host -> HBM upload time of the bench shard by copy-thread count (SAT_EXP_UPLOAD_THREADS)."""
import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1:
    import numpy as np
    import cuda_satabsearch_amd as sat
    db = sat.synth.make_db(125000, 32)
    with sat.Searcher(0) as s:
        ts = []
        for _ in range(5):
            t = time.perf_counter(); s.upload(db); ts.append((time.perf_counter() - t) * 1e3)
        print(f"threads={os.environ.get('SAT_EXP_UPLOAD_THREADS')}: upload ms {['%.2f' % x for x in ts]}  ({db.tab.nbytes*5/1e6:.0f} MB)", flush=True)
else:
    for t in ("1", "2", "4", "8"):
        subprocess.run([sys.executable, __file__, "x"], env=dict(os.environ, SAT_EXP_UPLOAD_THREADS=t))

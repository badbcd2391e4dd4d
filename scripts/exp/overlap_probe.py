#!/usr/bin/env python3
"""scripts/exp/overlap_probe.py [entries] [order] - wall time of sat_db_upload_search (upload and first
search overlapped) on the bench shard for several piece counts, next to upload-then-search."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import cuda_satabsearch_amd as sat

n = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
order = int(sys.argv[2]) if len(sys.argv) > 2 else 32
db = sat.synth.make_db(n, order, order)
q = sat.synth.make_query(order)


def med(f, k=5):
    ts = []
    for _ in range(k):
        t = time.perf_counter()
        f()
        ts.append((time.perf_counter() - t) * 1e3)
    return float(np.median(ts)), min(ts)


def plain(s):
    s.upload(db)
    s.search_async(True, False, 128)
    s.sync()


with sat.Searcher(0) as s:
    s.set_query(*q, 0)
    plain(s)
    print("upload, then search: median %.2f ms  min %.2f ms" % med(lambda: plain(s)), flush=True)
    want, _ = s.results()
for pieces in (1, 2, 4, 6, 8, 12, 16, 32):
    os.environ["SAT_EXP_UPLOAD_PIECES"] = str(pieces)
    with sat.Searcher(0) as s:
        s.set_query(*q, 0)
        s.upload_search(db, True, False, 128)
        m, lo = med(lambda: s.upload_search(db, True, False, 128))
        got, _ = s.results()
        assert np.array_equal(got, want)
        print("overlapped, %2d pieces: median %.2f ms  min %.2f ms  -> %.2f M scorings/s" % (pieces, m, lo, n / m / 1e3), flush=True)

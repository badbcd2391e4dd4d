"""Run by tests/test_gpu_parity.py::test_every_move_passes_the_references_self_check in a process of its own, with
SAT_DEVICE_LIB pointing at tests/native/libsat_selfcheck.so (the device library built with -DSAT_DIAG_SELFCHECK): small
searches through every kernel family; after each, the library's counters say how many proposed moves were checked
against a full recomputation of the score (the reference's TESTING assertion, K.cu:1105-1134) and how many differed.
Prints one JSON line."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cuda_satabsearch_amd as sat          # noqa: E402
from cuda_satabsearch_amd import _native    # noqa: E402
import oracle_lib                           # noqa: E402

assert "selfcheck" in _native.DEVICE_LIB, _native.DEVICE_LIB
lib = _native.device_lib()
out = (C.c_ulonglong * 16)()
cases = []
db = sat.synth.make_db(40, 1, 111, sort=False, seed=int(os.environ.get("SELFCHECK_SEED", "7")))
rng = np.random.default_rng(3)
with sat.Searcher(0) as s:
    s.upload(db)
    for n1 in (5, 16, 24, 32, 50, 101):
        src = int(rng.choice(np.nonzero(db.orders >= min(n1, 100))[0]))
        q = sat.synth.planted_query(db, src, keep=min(1.0, n1 / int(db.orders[src])))
        for lorder, lsoln in ((True, False), (False, True), (True, True)):
            s.set_query(*q, 0)
            sc, mp, _ = s.search(lorder, lsoln, 70)           # 70 restarts: a partial second wave
            lib.sat_diag_counters(out)
            want, wmap, _ = oracle_lib.search(db, *q, lorder, lsoln, 70)
            cases.append({"n1": int(q[0].shape[0]), "lorder": lorder, "lsoln": lsoln, "checks": int(out[9]), "mismatches": int(out[8]),
                          "scores_equal_oracle": bool(np.array_equal(sc, want)) and (not lsoln or bool(np.array_equal(mp, wmap))),
                          "kernels": s.last_launch_info()})
print(json.dumps(cases))

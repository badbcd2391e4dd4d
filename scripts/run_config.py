"""Run one named BASELINE workload at its stated size and print its throughput
(python scripts/run_config.py c2|c3|c3_1m|c4|mixed|lorderf|n96 [reps]).  Under
`rocprofv3 --kernel-trace --stats` this is what profiles/r02_config_* are taken from: the stats
table lists every kernel instantiation the workload dispatches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cuda_satabsearch_amd as sat
from cuda_satabsearch_amd import workloads as w

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
lorder, lsoln, r = True, False, 128
t0 = time.time()
if name == "c2":
    db, r = w.config2_db(), 4096
    queries = [(t, d, ty) for _, t, d, ty in w.config2_queries()]
elif name == "c3":
    db, queries = sat.synth.make_db(125_000, 32), [w.config3_query()]
elif name == "c3_1m":
    db, queries = w.config3_db(), [w.config3_query()]
elif name == "c4":
    db, lsoln = w.config4_db(), True
    queries = [w.config4_query()[1:]]
elif name == "mixed":
    db, queries = w.mixed_db(), [w.config3_query()]
elif name == "lorderf":
    db, queries, lorder = sat.synth.make_db(40_000, 32), [w.config3_query()], False
elif name == "lorderf_lsoln":
    db, queries, lorder, lsoln = sat.synth.make_db(40_000, 32), [w.config3_query()], False, True
elif name == "n96":
    db, queries = sat.synth.make_db(20_000, 96), [w.config3_query()]
elif name == "n64":
    db, queries = sat.synth.make_db(40_000, 64), [w.config3_query()]
elif name == "q200":
    # the reference paper's workload shape (scripts/mkquery200tab.sh, *querylist*.sh): 200 database members
    # as a query list against a ~15 000-entry size-sorted database, one batch
    db = sat.synth.make_db(15_000, 4, 40, sort=True)
    pick = np.random.default_rng(5).choice(len(db), 200, replace=False)
    queries = [(*db.dense(int(s)), db.ssetypes(int(s))) for s in pick]
elif name == "q101":
    db, queries = sat.synth.make_db(20_000, 8, 96, sort=True), [w.config4_query()[1:]]
else:
    raise SystemExit("unknown workload " + name)
gen_s = time.time() - t0
with sat.Searcher(0) as s:
    t0 = time.time()
    s.upload(db)
    up_ms = (time.time() - t0) * 1e3
    s.set_queries(queries, 0)
    s.search_timed(lorder, lsoln, r, 1)
    tot, _ = s.search_timed(lorder, lsoln, r, reps)
    sc, _, _ = s.search(lorder, lsoln, r)
ms = tot / reps
nsc = len(db) * len(queries)
print(f"{name}: {len(queries)} quer{'y' if len(queries) == 1 else 'ies'} x {len(db)} entries (orders {db.orders.min()}..{db.orders.max()}), "
      f"lorder={lorder} lsoln={lsoln} r={r}: {ms:.2f} ms/search -> {nsc / ms * 1e3:,.0f} scorings/s, "
      f"{nsc * r * 100 / ms * 1e3 / 1e9:.1f} G steps/s  (generate {gen_s:.0f} s, upload {up_ms:.0f} ms, checksum {int(sc.astype(np.int64).sum())})",
      flush=True)

"""First GPU bring-up check (scratch): GPU vs oracle(PHILOX) on the 586-entry db."""
import gzip, os, shutil, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import cuda_satabsearch_amd as sat
import oracle_lib

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tmp = tempfile.mkdtemp()
with gzip.open(os.path.join(root, "tests/golden/inputs/tableauxdistmatrixdb.small.ascii.gz"), "rb") as fi, open(os.path.join(tmp, "db.ascii"), "wb") as fo:
    shutil.copyfileobj(fi, fo)
db = sat.StructSet.read(os.path.join(tmp, "db.ascii"))
print("db", len(db), db.orders.min(), db.orders.max())
s = sat.Searcher(0)
s.upload(db)
for qf, lorder, lsoln in [("d1ubia_.input", True, False), ("d2phlb1.input", True, True), ("d2phlb1.input", False, True), ("d1twfa_.input", True, True)]:
    qs = sat.StructSet.read(os.path.join(root, "tests/golden/inputs", qf), "query", skip_header_lines=2)
    t, d = qs.dense(0)
    types = qs.ssetypes(0)
    s.set_query(t, d, types, 0)
    t0 = time.time()
    sc, sm, ms = s.search(lorder, lsoln, 128)
    t1 = time.time()
    osc, osm, _ = oracle_lib.search(db, t, d, types, lorder, lsoln, 128)
    t2 = time.time()
    nbad = int((sc != osc).sum())
    mbad = int((sm != osm).any(axis=1).sum()) if lsoln else 0
    print(f"{qf} n1={len(types)} lorder={lorder} lsoln={lsoln}: gpu {ms:.2f} ms (wall {t1-t0:.3f}s) oracle {t2-t1:.2f}s score-mismatch {nbad}/{len(sc)} map-mismatch {mbad}")
    if nbad:
        bad = np.nonzero(sc != osc)[0][:10]
        print("  first bad:", bad, sc[bad], osc[bad], db.orders[bad])

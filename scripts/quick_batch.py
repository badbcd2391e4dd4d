"""Query batches: NQ queries (the first NQ db entries with 8..32 SSEs, i.e. the `-q` list mode) scored
against the reference's 586-entry example database in one search() - python scripts/quick_batch.py [NQ]"""
import gzip, os, shutil, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cuda_satabsearch_amd as sat
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 48
tmp = tempfile.mkdtemp()
with gzip.open(os.path.join(root, "tests/golden/inputs/tableauxdistmatrixdb.small.ascii.gz"), "rb") as fi, open(os.path.join(tmp, "db.ascii"), "wb") as fo:
    shutil.copyfileobj(fi, fo)
db = sat.StructSet.read(os.path.join(tmp, "db.ascii"))
picks = [s for s in range(len(db.orders)) if 8 <= int(db.orders[s]) <= 32][:nq]
queries = []
for s in picks:
    t, d = db.dense(s)
    queries.append((t, d, db.ssetypes(s)))
with sat.Searcher(0) as s:
    s.upload(db)
    s.set_queries(queries, 0)
    s.search_timed(True, False, 128, 1)
    tot, _ = s.search_timed(True, False, 128, 3)
    ms = tot / 3
    print(f"{len(queries)} queries (orders {min(int(db.orders[i]) for i in picks)}..{max(int(db.orders[i]) for i in picks)}) x {len(db.orders)} entries, r=128: "
          f"{ms:.3f} ms per batch -> {len(queries) * len(db.orders) / ms * 1e3:,.0f} scorings/s")

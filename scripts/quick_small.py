"""Latency of single-query searches over the reference's 586-entry example database."""
import gzip, os, shutil, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cuda_satabsearch_amd as sat
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tmp = tempfile.mkdtemp()
with gzip.open(os.path.join(root, "tests/golden/inputs/tableauxdistmatrixdb.small.ascii.gz"), "rb") as fi, open(os.path.join(tmp, "db.ascii"), "wb") as fo:
    shutil.copyfileobj(fi, fo)
db = sat.StructSet.read(os.path.join(tmp, "db.ascii"))
with sat.Searcher(0) as s:
    s.upload(db)
    for qf in ("d1ubia_.input", "d2phlb1.input", "d1twfa_.input"):
        qs = sat.StructSet.read(os.path.join(root, "tests/golden/inputs", qf), "query", skip_header_lines=2)
        s.set_query_from(qs, 0)
        s.search_timed(True, False, 128, 2)
        tot, _ = s.search_timed(True, False, 128, 10)
        print(f"{qf} n1={qs.orders[0]}: {tot/10:.3f} ms per search of 586 entries, r=128 -> {586/(tot/10)*1e3:,.0f} scorings/s  ({586*128*100/(tot/10)/1e3:,.0f} M iterations/s; reference A100: 5.931 ms, 1264.68 M it/s for the 8-SSE query)")

"""BASELINE.json configs[2], [3] and [4] at their STATED sizes (-m gpu), through the C ABI and the
command line.  The oracle cannot run 100 000+ entries in test time, so each config checks:
(a) a 48-entry random sample against the oracle bit for bit (scores, and maps where LSOLN),
(b) two runs identical, (c) planted / self hits on top, (d) the scoring function's bounds;
and prints scorings/s.  The instantiations these runs dispatch are the ones profiled under
profiles/r02_config*.
"""
import os
import subprocess

import numpy as np
import pytest

import cuda_satabsearch_amd as sat
from cuda_satabsearch_amd import workloads
import oracle_lib

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "cuda_satabsearch_amd", "bin", "satabsearch")


def score_bound(n1):
    return 2 * n1 * (n1 - 1) // 2


# ------------------------------------------------------------------------------------------ configs[2]
@pytest.fixture(scope="module")
def c2_db():
    return workloads.config2_db()


def test_config2_multiquery_100k_r4096(c2_db, golden_dir):
    """d2phlb1 + multiquery (19, 8, 13, 101 SSEs) x 100 000 size-sorted entries of 8..32 SSEs,
    r = 4096, as one query batch (fermi_qlist_gpucudaSaTabsearch.e1462446:19-52 is this job on the
    reference's database)."""
    queries = workloads.config2_queries(golden_dir)
    assert [len(q[3]) for q in queries] == [19, 8, 13, 101]
    n = len(c2_db)
    sample = workloads.sample_entries(n, 48)
    with sat.Searcher(0) as s:
        s.upload(c2_db)
        s.set_queries([(t, d, ty) for _, t, d, ty in queries], first_query_ordinal=0)
        a, _, ms = s.search(True, False, 4096)
        s.search_async(True, False, 4096)
        b, _ = s.results()
    assert a.shape == (4, n) and np.array_equal(a, b)
    for qi, (name, t, d, ty) in enumerate(queries):
        osc, _, _ = oracle_lib.search(c2_db, t, d, ty, True, False, 4096, entries=sample, query_ordinal=qi)
        assert np.array_equal(a[qi][sample], osc), f"query {name}"
        assert np.abs(a[qi]).max() <= score_bound(len(ty))
    print(f"\nconfigs[2] 4 queries x {n} entries, r=4096: {ms:.1f} ms -> {4 * n / ms * 1e3:,.0f} scorings/s, "
          f"{4 * n * 4096 * 100 / ms * 1e3 / 1e9:.1f} G steps/s")


def test_config2_query_list_mode_through_the_command_line(c2_db, tmp_path):
    """The `-q` form: SIDs on stdin (cut to 7 characters, matched case-insensitively), queries are
    members of the 100 000-entry database; LTYPE/LORDER/LSOLN fixed T T F.  Rows of a 48-entry
    sample are compared, text and all, with the oracle's scores under the same streams."""
    n = len(c2_db)
    c2_db.write_ascii(tmp_path / "db100k.ascii")
    picks = [n - 1, 40_000, 77_777, 12_345]               # 32, ~17, ~26, ~11 SSEs
    sids = "".join((c2_db.names[p].upper() if k % 2 else c2_db.names[p]) + "\n" for k, p in enumerate(picks)).encode()
    p = subprocess.run([CLI, "-r", "4096", "-q", "db100k.ascii"], input=sids, cwd=tmp_path, capture_output=True)
    assert p.returncode == 0, p.stderr.decode()[-500:]
    lines = p.stdout.decode().splitlines()
    assert len(lines) == 4 * (n + 3)
    sample = workloads.sample_entries(n, 48, seed=1)
    for qi, src in enumerate(picks):
        block = lines[qi * (n + 3):(qi + 1) * (n + 3)]
        assert block[:3] == sat.report.header_lines(c2_db.names[src], "db100k.ascii", True, True, False)
        t, d = c2_db.dense(src)
        ty = c2_db.ssetypes(src)
        osc, _, _ = oracle_lib.search(c2_db, t, d, ty, True, False, 4096, entries=sample, query_ordinal=qi)
        expect = sat.report.result_lines([c2_db.names[e] for e in sample], c2_db.orders[sample], osc, len(ty))
        assert [block[3 + e] for e in sample] == expect, f"query {qi}"
        scores = np.array([int(l.split()[1]) for l in block[3:]])
        assert scores[src] == scores.max()                 # a db member's best hit is itself
    gpu_ms = [float(l.split()[3]) for l in p.stderr.decode().splitlines() if l.startswith("GPU execution time")]
    print(f"\nconfigs[2] -q: 4 SIDs x {n} entries, r=4096: GPU execution time {sum(gpu_ms):.1f} ms "
          f"(includes the first-launch code load and the result download)")


# ------------------------------------------------------------------------------------------ configs[3]
def test_config3_one_million_entries_on_one_gpu():
    """32-SSE query x 1 000 000 32-SSE entries, r = 128: the whole configs[3] database on ONE MI355X
    (2.7 GB packed in HBM; bench.py times its 125 000-entry shards)."""
    n = 1_000_000
    db = workloads.config3_db(n)
    planted = 765_432
    q = sat.synth.planted_query(db, planted, keep=1.0, jitter=0.5)
    rq = workloads.config3_query()
    sample = workloads.sample_entries(n, 48, seed=3)
    with sat.Searcher(0) as s:
        s.upload(db)
        s.set_query(*q, 0)
        a, _, ms_p = s.search(True, False, 128)
        b, _, _ = s.search(True, False, 128)
        s.set_query(*rq, 0)
        r, _, ms = s.search(True, False, 128)
        top_idx, top_sc = s.topk(5)
    assert np.array_equal(a, b)
    assert a.argmax() == planted
    assert np.abs(a).max() <= score_bound(32) and np.abs(r).max() <= score_bound(32)
    osc, _, _ = oracle_lib.search(db, *q, True, False, 128, entries=sample)
    assert np.array_equal(a[sample], osc)
    osc, _, _ = oracle_lib.search(db, *rq, True, False, 128, entries=sample)
    assert np.array_equal(r[sample], osc)
    order = np.lexsort((np.arange(n), -r.astype(np.int64)))[:5]
    assert np.array_equal(top_idx, order) and np.array_equal(top_sc, r[order])
    print(f"\nconfigs[3] 32-SSE query x {n} entries on one GPU, r=128: {ms:.1f} ms -> {n / ms * 1e3:,.0f} scorings/s "
          f"(planted query: {n / ms_p * 1e3:,.0f})")


# ------------------------------------------------------------------------------------------ configs[4]
def test_config4_large_query_lsoln_100k(golden_dir):
    """d1twfa_ (101 SSEs, with its >= 100 A parse quirk) x 100 000 entries of 8..111 SSEs (C5, size
    sorted), LSOLN = T, r = 128: every order bucket, both bit-set widths, several lanes per chain for
    the largest entries, solution maps for every entry."""
    n = 100_000
    db = workloads.config4_db(n)
    assert db.orders.min() == 8 and db.orders.max() == 111 and 500 < (db.orders > 96).sum() < 1500
    name, t, d, ty = workloads.config4_query(golden_dir)
    assert len(ty) == 101
    # the sample covers every order bucket (sorted db: spread positions) and the large class
    sample = np.unique(np.concatenate([workloads.sample_entries(n, 40, seed=4), np.arange(n - 8, n)]))
    with sat.Searcher(0) as s:
        s.upload(db)
        s.set_query(t, d, ty, 0)
        a, amaps, ms = s.search(True, True, 128)
        b, bmaps, _ = s.search(True, True, 128)
    assert np.array_equal(a, b) and np.array_equal(amaps, bmaps)
    osc, omaps, _ = oracle_lib.search(db, t, d, ty, True, True, 128, entries=sample)
    assert np.array_equal(a[sample], osc)
    assert np.array_equal(amaps[sample], omaps)
    # maps are injective, order preserving and inside the entry
    for e in sample:
        m = amaps[e][:101]
        img = m[m >= 0]
        assert (np.diff(img) > 0).all() and (img < db.orders[e]).all()
    assert (amaps[:, 101:] == -1).all()
    print(f"\nconfigs[4] 101-SSE query x {n} entries (8..111 SSEs), LSOLN=T, r=128: {ms:.1f} ms -> {n / ms * 1e3:,.0f} scorings/s")


def test_motif_workflow_without_order_constraint_r4096(c2_db, golden_dir):
    """the substructure-motif workflow (scripts/qptabmatchstructs.sh:135-137, 156 runs
    `T F T` with -r4096): the reference's 9-SSE sheet motif and an 8-SSE non-sequential motif cut out of a
    database member, LORDER = F, LSOLN = T, r = 4096, against 100 000 entries - the [0, n2) candidate window
    (K.cu:1079-1083) and solution maps at scale.  Sampled entries equal the oracle bit for bit, maps are
    injective and inside the entry but need not be ordered, and the member the motif was cut from is on top."""
    n = len(c2_db)
    name, t, d, ty = workloads.load_queries("1qlp_sheetbc.input", golden_dir)[0]
    src = 91_234
    st, sd = c2_db.dense(src)
    pick = np.random.default_rng(9).permutation(int(c2_db.orders[src]))[:8]      # not in sequence order
    cut = (st[np.ix_(pick, pick)].copy(), sd[np.ix_(pick, pick)].copy(), np.diagonal(st)[pick].copy())
    sample = workloads.sample_entries(n, 16, seed=6)
    with sat.Searcher(0) as s:
        s.upload(c2_db)
        s.set_queries([(t, d, ty), cut], 0)
        a, amaps, ms = s.search(False, True, 4096)
        b, bmaps, _ = s.search(False, True, 4096)
    assert np.array_equal(a, b) and np.array_equal(amaps, bmaps)
    unordered = 0
    for qi, q in enumerate([(t, d, ty), cut]):
        n1 = len(q[2])
        osc, omaps, _ = oracle_lib.search(c2_db, *q, False, True, 4096, entries=sample, query_ordinal=qi)
        assert np.array_equal(a[qi][sample], osc), f"query {qi}"
        assert np.array_equal(amaps[qi][sample], omaps), f"maps of query {qi}"
        assert np.abs(a[qi]).max() <= score_bound(n1)
        for e in sample:
            img = amaps[qi][e][:n1]
            img = img[img >= 0]
            assert len(set(img.tolist())) == len(img) and (img < c2_db.orders[e]).all()
            unordered += int((np.diff(img) < 0).any())
        assert (amaps[qi][:, n1:] == -1).all()
    assert unordered > 0                                   # the order constraint really is off
    assert a[1][src] == a[1].max() == score_bound(8)       # the cut motif matches its source exactly
    assert np.array_equal(np.sort(amaps[1][src][:8]), np.sort(pick))
    print(f"\nmotif workflow: 2 motifs (9, 8 SSEs) x {n} entries, T F T, r=4096: {ms:.1f} ms -> {2 * n / ms * 1e3:,.0f} scorings/s")


def test_mixed_size_database_throughput_and_stream_overlap(monkeypatch):
    """A size-sorted database with orders uniform on [8, 32] (C3): the order buckets of a search
    run concurrently on side streams; queueing them one after the other instead must give the same
    scores.  Prints both rates."""
    db = workloads.mixed_db(100_000)
    q = workloads.config3_query()
    sample = workloads.sample_entries(len(db), 48, seed=5)
    rates = {}
    ref = None
    for streams in ("1", "0"):
        monkeypatch.setenv("SAT_EXP_STREAMS", streams)
        with sat.Searcher(0) as s:
            s.upload(db)
            s.set_query(*q, 0)
            s.search_timed(True, False, 128, 1)
            tot, _ = s.search_timed(True, False, 128, 3)
            sc, _, _ = s.search(True, False, 128)
        rates[streams] = len(db) / (tot / 3) * 1e3
        if ref is None:
            ref = sc
            osc, _, _ = oracle_lib.search(db, *q, True, False, 128, entries=sample)
            assert np.array_equal(sc[sample], osc)
        else:
            assert np.array_equal(sc, ref)
    print(f"\nmixed-size db (orders 8..32, sorted) x 32-SSE query, r=128: {rates['1']:,.0f} scorings/s with the buckets "
          f"on side streams, {rates['0']:,.0f} queued on one stream")


def test_query_list_workload_of_the_paper():
    """The reference's published workload shape (scripts/mkquery200tab.sh, *querylist*.sh): 200 database
    members as ONE query batch against a ~15 000-entry size-sorted database, r = 128: every query's own
    entry is (or ties) its best hit, a sample of (query, entry) pairs equals the oracle bit for bit, and the
    best-10 rows per query come back without the 12 MB of score arrays."""
    db = sat.synth.make_db(15_000, 4, 40, sort=True)
    pick = np.random.default_rng(5).choice(len(db), 200, replace=False)
    queries = [(*db.dense(int(s)), db.ssetypes(int(s))) for s in pick]
    with sat.Searcher(0) as s:
        s.upload(db)
        s.set_queries(queries, 0)
        scores, _, ms = s.search(True, False, 128)
        before = s.d2h_bytes()
        hits = s.topk_hits(10)
        assert s.d2h_bytes() - before == 200 * 10 * 32
    assert scores.shape == (200, len(db))
    for qi, src in enumerate(pick):
        assert scores[qi][src] == scores[qi].max()
        order = np.lexsort((np.arange(len(db)), -scores[qi].astype(np.int64)))[:10]
        assert np.array_equal(hits[qi]["entry"], order)
    sample = workloads.sample_entries(len(db), 24, seed=8)
    for qi in (0, 57, 199):
        osc, _, _ = oracle_lib.search(db, *queries[qi], True, False, 128, entries=sample, query_ordinal=qi)
        assert np.array_equal(scores[qi][sample], osc)
    print(f"\n200 queries x {len(db)} entries, r=128, one batch: {ms:.1f} ms -> {200 * len(db) / ms * 1e3:,.0f} scorings/s")

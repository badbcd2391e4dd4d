"""T3: a parallel run cannot replay the reference's ONE sequential drand48 stream, so against
the reference's golden `-c` stdout a result must look like one more SEED of the same algorithm.  What that means is
measured (tests/golden/make_seed_spread.py -> expected/seed_spread.json: every pair of eight drand48 seeds of the `-c`
semantics compared entry by entry on five jobs); here a run's figures against the golden output are checked against
those bands.  Shared by the GPU test (the kernel) and the CPU test (the oracle on the kernel's Philox streams: the
same statement, since the two are bit-identical)."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# job -> (stdin file of tests/golden/inputs, golden stdout, restarts, lorder, lsoln)
JOBS = {
    "c1_d1ubia_small.r128": ("c1_d1ubia_small.input", "c1_d1ubia_small.r128.out", 128, True, False),
    "d2phlb1.r4096": ("d2phlb1.input", "recorded_2013_d2phlb1.r4096.out", 4096, True, False),
    "d1twfa_.r128": ("d1twfa_.input", "d1twfa_.r128.out", 128, True, False),
    "d2phlb1_TFT.r128": ("d2phlb1_TFT.input", "d2phlb1_TFT.r128.out", 128, False, True),
    "multiquery.r128": ("multiquery.input", "multiquery.r128.out", 128, True, False),
}


def parse_rows(stdout):
    """{(query block, name): (score, ((query SSE, db SSE), ...))}: see tests/golden/make_seed_spread.py"""
    rows, qids, q, last = {}, [], -1, None
    for l in stdout.splitlines():
        if l.startswith("# QUERY ID"):
            qid = l.split("=", 1)[1].strip()
            if qid not in qids:
                qids.append(qid)
            q = qids.index(qid)
        if not l or l.startswith("#"):
            continue
        t = l.split()
        if len(t) == 5:
            last = (q, t[0])
            rows[last] = (int(t[1]), ())
        elif len(t) == 2 and last is not None:
            rows[last] = (rows[last][0], rows[last][1] + ((int(t[0]), int(t[1])),))
    return rows


def map_pairs(ssemap_row):
    """int map[i] = db SSE or -1  ->  the 1-based pairs the reference prints (cudaSaTabsearch.cu:448-453)"""
    return tuple((i + 1, int(j) + 1) for i, j in enumerate(ssemap_row) if j >= 0)


def check(job, names, run_scores, run_maps=None):
    """run_scores[q][e] (and run_maps[q][e] = tuple of pairs) of the run under test, db order `names`."""
    spread = json.load(open(os.path.join(ROOT, "tests/golden/expected/seed_spread.json")))["jobs"][job]
    band = spread["band"]
    gold = parse_rows(open(os.path.join(ROOT, "tests/golden/expected", JOBS[job][1])).read())
    nq = len(run_scores)
    assert len(gold) == nq * len(names) == spread["entries"]
    ref = np.array([gold[(q, n)][0] for q in range(nq) for n in names])
    got = np.concatenate([np.asarray(run_scores[q]) for q in range(nq)])
    diff = got - ref
    frac, mx, mean = float((diff != 0).mean()), int(np.abs(diff).max()), float(diff.mean())
    rc = float(np.corrcoef(np.argsort(np.argsort(got)), np.argsort(np.argsort(ref)))[0, 1])
    print(f"\n{job}: run vs golden: {frac:.3f} of entries differ (band {band['frac_differing']}), max |diff| {mx} "
          f"(band {band['max_abs']}), mean {mean:+.3f} (|band| {band['abs_mean']}), rank corr {rc:.4f} (band {band['rank_corr']})")
    assert band["frac_differing"][0] - 0.03 <= frac <= band["frac_differing"][1] + 0.03
    assert mx <= band["max_abs"][1] + 2
    assert abs(mean) <= band["abs_mean"][1] + 0.05
    assert rc >= band["rank_corr"][0] - 0.01
    if run_maps is not None:
        same = np.array([run_maps[q][e] == gold[(q, n)][1] for q in range(nq) for e, n in enumerate(names)])
        fs, fss = float(same.mean()), float(same[diff == 0].mean())
        print(f"{job}: identical solution maps on {fs:.3f} of the entries (band {band['frac_same_map']}), on {fss:.3f} of "
              f"those with equal scores (band {band['frac_same_map_given_same_score']})")
        assert band["frac_same_map"][0] - 0.03 <= fs <= band["frac_same_map"][1] + 0.03
        assert band["frac_same_map_given_same_score"][0] - 0.06 <= fss <= band["frac_same_map_given_same_score"][1] + 0.06

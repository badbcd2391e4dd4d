#!/usr/bin/env python3
"""tests/golden/make_seed_spread.py - measure the seed-to-seed spread of the reference's `-c`
semantics and write tests/golden/expected/seed_spread.json.

The reference's host run draws from ONE sequential drand48 stream seeded srand48(1234)
(cudaSaTabsearch.cu:871); no parallel run can replay it, so a GPU result can only be compared with
its golden stdout statistically.  What "statistically" means is measured
here: the oracle CLI (pinned byte for byte to the reference's `-c` output at seed 1234) is run with
other drand48 seeds (-S) on the two jobs the T3 test uses, and every pair of runs is compared
entry by entry: fraction of entries whose score differs, largest |difference|, mean difference.
The committed JSON holds the per-pair figures and their ranges; the GPU test requires the
GPU-vs-golden figures to lie inside those ranges (tests/test_gpu_parity.py).

Runs in the build container only (about one CPU-minute per seed for the r=4096 job); needs
oracle/oracle_cli (make -C oracle) and the unpacked golden inputs.
"""
import gzip
import itertools
import json
import os
import shutil
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CLI = os.path.join(ROOT, "oracle", "oracle_cli")
SEEDS = [1234, 1, 2, 3, 5, 8, 13, 21]          # 1234 = the reference's own seed

JOBS = {
    # name: (stdin file, extra args, golden stdout that seed 1234 must reproduce)
    "c1_d1ubia_small.r128": ("c1_d1ubia_small.input", ["-r", "128"], "c1_d1ubia_small.r128.out"),
    # the 2013 recording was made with MAXDIM_GPU = 32 (two passes split at 32 SSEs): -m 32
    "d2phlb1.r4096": ("d2phlb1.input", ["-r", "4096", "-m", "32"], "recorded_2013_d2phlb1.r4096.out"),
    # the large-query path (101 SSEs), the no-order path with solution maps, and a three-query stream
    "d1twfa_.r128": ("d1twfa_.input", ["-r", "128"], "d1twfa_.r128.out"),
    "d2phlb1_TFT.r128": ("d2phlb1_TFT.input", ["-r", "128"], "d2phlb1_TFT.r128.out"),
    "multiquery.r128": ("multiquery.input", ["-r", "128"], "multiquery.r128.out"),
}


def parse_rows(stdout):
    """{(query block, name): (score, ((query SSE, db SSE), ...))} of a cudaSaTabsearch stdout: a block of three '#'
    header lines per (query, size class), rows `name score norm2 z p`, with LSOLN the 1-based pairs of the
    row's solution map on the lines after it.  Query blocks are numbered by QUERY ID in order of appearance
    (the two size classes of one query share the number)."""
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


def scores_by_name(stdout):
    return {k: v[0] for k, v in parse_rows(stdout).items()}


def main():
    work = tempfile.mkdtemp()
    for f in os.listdir(os.path.join(HERE, "inputs")):
        src = os.path.join(HERE, "inputs", f)
        if f.endswith(".gz"):
            with gzip.open(src, "rb") as fi, open(os.path.join(work, f[:-3]), "wb") as fo:
                shutil.copyfileobj(fi, fo)
        else:
            shutil.copy(src, os.path.join(work, f))

    def run(job, seed):
        stdin_file, args, _ = JOBS[job]
        with open(os.path.join(work, stdin_file), "rb") as fin:
            p = subprocess.run([CLI, "-c", "-S", str(seed)] + args, stdin=fin, cwd=work, capture_output=True)
        assert p.returncode == 0, p.stderr.decode()[-300:]
        return p.stdout.decode()

    out = {"seeds": SEEDS, "note": "oracle_cli -c -S <seed>; pairwise over all seeds; 'vs_1234' rows are the "
           "pairs that include the reference's own seed", "jobs": {}}
    with ThreadPoolExecutor(max_workers=int(os.environ.get("JOBS", "8"))) as ex:
        futs = {(job, seed): ex.submit(run, job, seed) for job in JOBS for seed in SEEDS}
        res = {k: f.result() for k, f in futs.items()}
    for job, (_, _, golden) in JOBS.items():
        gold = open(os.path.join(HERE, "expected", golden)).read()
        assert scores_by_name(res[(job, 1234)]) == scores_by_name(gold), f"{job}: seed 1234 does not reproduce {golden}"
        names = sorted(scores_by_name(gold))
        parsed = {s: parse_rows(res[(job, s)]) for s in SEEDS}
        vec = {s: np.array([parsed[s][n][0] for n in names]) for s in SEEDS}
        with_maps = any(parse_rows(gold)[n][1] for n in names)
        pairs = []
        for a, b in itertools.combinations(SEEDS, 2):
            d = vec[b] - vec[a]
            ra, rb = np.argsort(np.argsort(vec[a])), np.argsort(np.argsort(vec[b]))
            pairs.append({"seeds": [a, b], "frac_differing": float((d != 0).mean()), "max_abs": int(np.abs(d).max()),
                          "mean": float(d.mean()), "rank_corr": float(np.corrcoef(ra, rb)[0, 1])})
            if with_maps:
                # LSOLN: how often two seeds report the very same solution map, over all entries and over
                # the entries on which they agree about the score
                same_map = np.array([parsed[a][n][1] == parsed[b][n][1] for n in names])
                pairs[-1]["frac_same_map"] = float(same_map.mean())
                pairs[-1]["frac_same_map_given_same_score"] = float(same_map[d == 0].mean())
        rng = lambda key, f=lambda x: x: [min(f(p[key]) for p in pairs), max(f(p[key]) for p in pairs)]
        out["jobs"][job] = {
            "entries": len(names), "pairs": pairs,
            "band": {"frac_differing": rng("frac_differing"), "max_abs": rng("max_abs"),
                     "abs_mean": rng("mean", abs), "rank_corr": rng("rank_corr")},
        }
        if with_maps:
            out["jobs"][job]["band"]["frac_same_map"] = rng("frac_same_map")
            out["jobs"][job]["band"]["frac_same_map_given_same_score"] = rng("frac_same_map_given_same_score")
        print(job, json.dumps(out["jobs"][job]["band"]))
    with open(os.path.join(HERE, "expected", "seed_spread.json"), "w") as f:
        json.dump(out, f, indent=1)
    shutil.rmtree(work)


if __name__ == "__main__":
    sys.exit(main())

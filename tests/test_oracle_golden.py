"""The CPU oracle against the reference's own outputs (not gpu).

expected/*.out were produced by the reference's sources compiled in place
(oracle/_ref, tests/golden/make_golden.sh); recorded_2013_* is the stdout the reference
ships from a real 2013 run.  Byte-for-byte equality pins search + reader + statistics +
printing of the restatement.
"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "oracle", "oracle_cli")
EXPECTED = os.path.join(ROOT, "tests", "golden", "expected")


def run_cli(golden_dir, input_name, *args):
    with open(os.path.join(golden_dir, input_name)) as fin:
        p = subprocess.run([CLI, "-c", *args], stdin=fin, cwd=golden_dir, capture_output=True)
    assert p.returncode == 0, p.stderr.decode()[-500:]
    return p.stdout


CASES = [
    ("d1ubia_.input", "d1ubia_.r128.out", ["-r", "128"]),                  # 1-entry db, T T T
    ("d1ae6h1.input", "d1ae6h1.r128.out", ["-r", "128"]),
    ("d2phlb1.input2", "d2phlb12.r128.out", ["-r", "128"]),                # LSOLN=T solution pairs
    ("d2phlb1.input3", "d2phlb13.r128.out", ["-r", "128"]),
    ("1qlp_sheetbc.input", "1qlp_sheetbc.r128.out", ["-r", "128"]),        # substructure motif query
    ("c1_d1ubia_small.input", "c1_d1ubia_small.r128.out", ["-r", "128"]),  # BASELINE configs[0]
    ("d2phlb1.input", "d2phlb1.r128.out", ["-r", "128"]),
    ("d2phlb1_TFT.input", "d2phlb1_TFT.r128.out", ["-r", "128"]),          # LORDER=F, LSOLN=T
    ("d2phlb1_TTT.input", "d2phlb1_TTT.r128.out", ["-r", "128"]),
    ("d1twfa_.input", "d1twfa_.r128.out", ["-r", "128"]),                  # 101 SSEs, >=100 A parse quirk
    ("d1twfa_.input", "d1twfa_.r16.out", ["-r", "16"]),
    ("multiquery.input", "multiquery.r128.out", ["-r", "128"]),            # one stream across 3 queries
    ("readme_1ubq.input", "readme_1ubq.r128.out", ["-r", "128"]),          # the README's worked example (query body)
]


@pytest.mark.parametrize("inp,exp,args", CASES, ids=[c[1] for c in CASES])
def test_oracle_cli_matches_reference_build(golden_dir, inp, exp, args):
    out = run_cli(golden_dir, inp, *args)
    with open(os.path.join(EXPECTED, exp), "rb") as f:
        assert out == f.read()


def test_oracle_step_trace_matches_reference_debug_build(golden_dir):
    """Every SA step's proposal (ssei, startj, endj, newj) and map, one restart."""
    out = run_cli(golden_dir, "d1ubia_.input", "-r", "1", "-t")
    with open(os.path.join(EXPECTED, "d1ubia_.r1.trace.stdout"), "rb") as f:
        assert out == f.read()


def test_oracle_reproduces_recorded_2013_run(golden_dir):
    """old/nvcc_src_cuda5/cpu_cudaSaTabsearch.o1462445: `-c -r4096 < d2phlb1.input` on
    the 2013 sources, whose small/large class boundary was 32 SSEs (-m 32).  ~30 s."""
    out = run_cli(golden_dir, "d2phlb1.input", "-r", "4096", "-m", "32")
    with open(os.path.join(EXPECTED, "recorded_2013_d2phlb1.r4096.out"), "rb") as f:
        assert out == f.read()


def test_q_mode_matches_reference_build(golden_dir):
    """`-q dbfile` with SIDs on stdin: the golden stdout comes from the reference's own parser, kernel
    and statistics under oracle/ref_driver.cpp's replay of main's -q branch (SID cut to 7 characters,
    case-insensitive lookup, the query copied out of the database arrays off-diagonal cell by cell,
    cudaSaTabsearch.cu:631-664, 746-780, 356-400)."""
    with open(os.path.join(golden_dir, "qmode_sids.txt"), "rb") as f:
        sids = f.read()
    assert b"D1NLDL1xyz" in sids
    p = subprocess.run([CLI, "-c", "-r", "16", "-q", "tableauxdistmatrixdb.small.ascii"], input=sids, cwd=golden_dir,
                       capture_output=True)
    assert p.returncode == 0, p.stderr.decode()[-300:]
    with open(os.path.join(EXPECTED, "qmode_small.r16.out"), "rb") as f:
        gold = f.read()
    assert p.stdout == gold and gold.count(b"# QUERY ID") == 3 and b"# QUERY ID = d1nldl1" in gold


def test_q_mode_equals_inline_query(golden_dir):
    """-q takes SIDs of db members; the same structure given inline must give the same
    rows (options are fixed T T F in -q mode, cudaSaTabsearch.cu:633-635)."""
    import cuda_satabsearch_amd as sat
    db = sat.StructSet.read(os.path.join(golden_dir, "tableauxdistmatrixdb.small.ascii"))
    sid = db.names[3]
    p = subprocess.run([CLI, "-c", "-r", "8", "-q", "tableauxdistmatrixdb.small.ascii"],
                       input=(sid.upper() + "\n").encode(), cwd=golden_dir, capture_output=True)
    assert p.returncode == 0, p.stderr.decode()[-300:]
    sub = db.subset([3])
    inline = os.path.join(golden_dir, "_inline.input")
    body = os.path.join(golden_dir, "_inline.body")
    sat.synth.write_ascii(sub, body)
    with open(inline, "w") as f:
        f.write("tableauxdistmatrixdb.small.ascii\nT T F\n" + open(body).read())
    out2 = run_cli(golden_dir, "_inline.input", "-r", "8")
    assert p.stdout == out2


def test_readme_printed_rows_are_a_stale_vector(golden_dir):
    """README_example_usage.txt:43-49 prints the first seven rows of `-c < ubiquiin.query` (the 1UBQ
    query of :10-27 against the 586-entry example database): scores 6 4 2 4 8 5 7.  The current
    sources compiled here give 11 5 2 3 10 8 8 for the same input (readme_1ubq.r128.out, reproduced
    byte for byte by the oracle above), and so do five other drand48 seeds on the first four rows
    (11 5 2 3): the README's rows are from an older build and are NOT a parity target.  Recorded here
    so that the mismatch is a documented fact rather than a surprise."""
    printed = open(os.path.join(EXPECTED, "readme_1ubq.printed_head.txt")).read().splitlines()
    current = open(os.path.join(EXPECTED, "readme_1ubq.r128.out")).read().splitlines()[3:10]
    assert [l.split()[0] for l in printed] == [l.split()[0] for l in current]       # same entries, same order
    assert [int(l.split()[1]) for l in printed] == [6, 4, 2, 4, 8, 5, 7]
    assert [int(l.split()[1]) for l in current] == [11, 5, 2, 3, 10, 8, 8]
    for seed in ("1", "2", "3"):
        rows = run_cli(golden_dir, "readme_1ubq.input", "-r", "128", "-S", seed).decode().splitlines()[3:7]
        assert [int(l.split()[1]) for l in rows] == [11, 5, 2, 3]


# ---------------------------------------------------------------- T3 on the CPU: the kernel's streams against `-c`
@pytest.fixture(scope="module")
def small_db_cpu(golden_dir):
    import cuda_satabsearch_amd as sat
    return sat.StructSet.read(os.path.join(golden_dir, "tableauxdistmatrixdb.small.ascii"))


@pytest.mark.parametrize("job", ["c1_d1ubia_small.r128", "d1twfa_.r128", "d2phlb1_TFT.r128", "multiquery.r128"])
def test_philox_streams_look_like_one_more_seed_of_the_reference(golden_dir, small_db_cpu, job):
    """The statement of the GPU's T3 test without a GPU: the oracle on the KERNEL's random streams (Philox, 16-bit
    index draws - bit-identical to the kernel, tests/test_gpu_parity.py) against the reference's golden `-c` stdout
    lies inside the measured seed-to-seed spread of the reference's own stream (tests/t3_band.py), for the 8-SSE
    query, the 101-SSE query, LORDER = F with solution maps and the three-query stream.  (The r = 4096 job is left to
    the GPU test: 30 s of oracle time.)"""
    import cuda_satabsearch_amd as sat
    import oracle_lib
    import t3_band
    stdin_file, _, restarts, lorder, lsoln = t3_band.JOBS[job]
    qs = sat.StructSet.read(os.path.join(golden_dir, stdin_file), "query", skip_header_lines=2)
    scores, maps = [], []
    for k in range(len(qs)):
        t, d = qs.dense(k)
        sc, mp, _ = oracle_lib.search(small_db_cpu, t, d, qs.ssetypes(k), lorder, lsoln, restarts, query_ordinal=k)
        scores.append(sc)
        if lsoln:
            maps.append([t3_band.map_pairs(mp[e]) for e in range(len(small_db_cpu))])
    t3_band.check(job, small_db_cpu.names, scores, maps if lsoln else None)


def test_sixteen_bit_index_draws_against_full_resolution_draws(golden_dir, small_db_cpu):
    """The kernel picks the moved SSE and the candidate with 16-bit draws, (v16 + 1) * 2^-16: for
    n up to 111 the bins hold 590 or 591 of the 65536 values - at most 0.17 % apart - where the reference draws
    float uniforms at full resolution (curand_uniform / drand48).  A distribution change, however small: the oracle's
    comparison mode SA_RNG_PHILOX32 (a whole word per index draw) against the shipped layout must differ by no more
    than two seeds of the reference's own stream do, on the 8-SSE and the 101-SSE query."""
    import json
    import numpy as np
    import cuda_satabsearch_amd as sat
    import oracle_lib
    spread = json.load(open(os.path.join(ROOT, "tests/golden/expected/seed_spread.json")))["jobs"]
    for job, inp in (("c1_d1ubia_small.r128", "c1_d1ubia_small.input"), ("d1twfa_.r128", "d1twfa_.input")):
        qs = sat.StructSet.read(os.path.join(golden_dir, inp), "query", skip_header_lines=2)
        t, d = qs.dense(0)
        a, _, _ = oracle_lib.search(small_db_cpu, t, d, qs.ssetypes(0), True, False, 128)
        b, _, _ = oracle_lib.search(small_db_cpu, t, d, qs.ssetypes(0), True, False, 128, mode=2)
        band = spread[job]["band"]
        diff = b - a
        frac, mx, mean = float((diff != 0).mean()), int(np.abs(diff).max()), float(diff.mean())
        print(f"\n{job}: 16-bit vs 32-bit index draws: {frac:.3f} differ, max {mx}, mean {mean:+.3f}; seed-to-seed band {band}")
        assert frac <= band["frac_differing"][1] + 0.03 and mx <= band["max_abs"][1] + 2
        assert abs(mean) <= band["abs_mean"][1] + 0.05

"""The `satabsearch` command line (csrc/host/sat_main.c): same stdin / stdout surface as
the reference's cudaSaTabsearch.  Host mode (-c) is byte-identical to the reference's
-c output (not gpu); GPU mode is byte-identical to the oracle CLI running the same
Philox streams (gpu)."""
import os
import subprocess

import numpy as np
import pytest

import cuda_satabsearch_amd as sat

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "cuda_satabsearch_amd", "bin", "satabsearch")
ORACLE_CLI = os.path.join(ROOT, "oracle", "oracle_cli")
EXPECTED = os.path.join(ROOT, "tests", "golden", "expected")


def run(binary, cwd, args, stdin_path=None, stdin_bytes=None):
    if stdin_path:
        with open(os.path.join(cwd, stdin_path), "rb") as f:
            stdin_bytes = f.read()
    return subprocess.run([binary, *args], input=stdin_bytes, cwd=cwd, capture_output=True)


@pytest.mark.parametrize("name", ["c1_d1ubia_small", "d2phlb1_TFT", "d2phlb1_TTT", "multiquery", "d1twfa_"])
def test_host_mode_is_byte_identical_to_reference(golden_dir, name):
    p = run(CLI, golden_dir, ["-c", "-r", "128"], stdin_path=name + ".input")
    assert p.returncode == 0, p.stderr.decode()[-400:]
    assert p.stdout == open(os.path.join(EXPECTED, name + ".r128.out"), "rb").read()


def test_host_mode_query_list_matches_reference_build(golden_dir):
    """The product's `-c -q` against the reference-built golden (tests/golden/make_golden.sh)."""
    sids = open(os.path.join(golden_dir, "qmode_sids.txt"), "rb").read()
    p = run(CLI, golden_dir, ["-c", "-r", "16", "-q", "tableauxdistmatrixdb.small.ascii"], stdin_bytes=sids)
    assert p.returncode == 0, p.stderr.decode()[-300:]
    assert p.stdout == open(os.path.join(EXPECTED, "qmode_small.r16.out"), "rb").read()


def test_host_mode_query_list(golden_dir):
    sids = b"d1kcul1\nD1NLDL1\n"
    a = run(CLI, golden_dir, ["-c", "-r", "16", "-q", "tableauxdistmatrixdb.small.ascii"], stdin_bytes=sids)
    b = run(ORACLE_CLI, golden_dir, ["-c", "-r", "16", "-q", "tableauxdistmatrixdb.small.ascii"], stdin_bytes=sids)
    assert a.returncode == 0 and b.returncode == 0
    assert a.stdout == b.stdout and a.stdout.count(b"# QUERY ID") == 2


def test_binary_image_cache_gives_identical_output(golden_dir, tmp_path):
    import shutil
    for f in ("tableauxdistmatrixdb.small.ascii", "c1_d1ubia_small.input"):
        shutil.copy(os.path.join(golden_dir, f), tmp_path / f)
    first = run(CLI, str(tmp_path), ["-c", "-r", "8", "-b"], stdin_path="c1_d1ubia_small.input")
    assert first.returncode == 0 and (tmp_path / "tableauxdistmatrixdb.small.ascii.satbin").exists()
    again = run(CLI, str(tmp_path), ["-c", "-r", "8", "-b"], stdin_path="c1_d1ubia_small.input")
    plain = run(CLI, str(tmp_path), ["-c", "-r", "8"], stdin_path="c1_d1ubia_small.input")
    assert b"binary image" in again.stderr and b"binary image" not in first.stderr
    assert first.stdout == again.stdout == plain.stdout


def test_input_errors(golden_dir):
    p = run(CLI, golden_dir, ["-c"], stdin_bytes=b"nosuchfile.ascii\nT T F\n" + open(os.path.join(golden_dir, "d1ubia_.input"), "rb").read().split(b"\n", 2)[2])
    assert p.returncode == 1 and b"ERROR opening db file" in p.stderr
    p = run(CLI, golden_dir, ["-c", "-q", "tableauxdistmatrixdb.small.ascii"], stdin_bytes=b"nosuchsid\n")
    assert p.returncode == 1 and b"not found" in p.stderr
    p = run(CLI, golden_dir, ["-c"], stdin_bytes=b"")
    assert p.returncode == 1


def test_gpu_mode_without_device_is_an_error(golden_dir):
    if sat.device_count() > 0:
        pytest.skip("a GPU is present")
    p = run(CLI, golden_dir, ["-r", "8"], stdin_path="d1ubia_.input")
    assert p.returncode == 1 and b"no usable HIP device" in p.stderr and p.stdout.count(b"\n") <= 3


@pytest.mark.gpu
@pytest.mark.parametrize("name,restarts", [("c1_d1ubia_small", 128), ("d2phlb1_TTT", 128), ("d2phlb1_TFT", 64),
                                           ("multiquery", 128), ("d1ubia_", 128)])
def test_gpu_mode_matches_oracle_cli(golden_dir, name, restarts):
    a = run(CLI, golden_dir, ["-r", str(restarts)], stdin_path=name + ".input")
    b = run(ORACLE_CLI, golden_dir, ["-c", "-p", "-G", "-r", str(restarts)], stdin_path=name + ".input")
    assert a.returncode == 0, a.stderr.decode()[-400:]
    assert b.returncode == 0
    assert a.stdout == b.stdout


@pytest.mark.gpu
def test_gpu_mode_query_list_and_large_class(tmp_path):
    """-q mode over a database that has both size classes: large-class blocks come after
    every query's small-class block, with the reference GPU path's row format."""
    db = sat.synth.make_db(60, 70, 111, sort=False, seed=11)
    assert (db.orders > 96).sum() > 3 and (db.orders <= 96).sum() > 3
    db.names = ["m%06d" % i for i in range(len(db))]        # -q cuts SIDs to 7 characters
    sat.synth.write_ascii(db, tmp_path / "mix.ascii")
    sids = (db.names[5] + "\n" + db.names[int(np.argmax(db.orders > 96))].upper() + "\n").encode()
    a = run(CLI, str(tmp_path), ["-r", "64", "-q", "mix.ascii"], stdin_bytes=sids)
    b = run(ORACLE_CLI, str(tmp_path), ["-c", "-p", "-G", "-r", "64", "-q", "mix.ascii"], stdin_bytes=sids)
    assert a.returncode == 0, a.stderr.decode()[-400:]
    assert a.stdout == b.stdout
    assert a.stdout.count(b"# QUERY ID") == 4


@pytest.mark.gpu
def test_gpu_mode_topk_is_sorted_head_of_full_output(golden_dir):
    full = run(CLI, golden_dir, ["-r", "128"], stdin_path="multiquery.input")
    top = run(CLI, golden_dir, ["-r", "128", "-k", "7"], stdin_path="multiquery.input")
    assert full.returncode == 0 and top.returncode == 0, top.stderr.decode()[-300:]
    blocks, cur = [], None
    for line in full.stdout.decode().splitlines():
        if line.startswith("# cudaSaTabsearch"):
            cur = {"head": [line], "rows": []}
            blocks.append(cur)
        elif line.startswith("#"):
            cur["head"].append(line)
        else:
            cur["rows"].append(line)
    expect = []
    for b in blocks:
        rows = sorted(enumerate(b["rows"]), key=lambda t: (-int(t[1].split()[1]), t[0]))[:7]
        expect += b["head"] + [r for _, r in rows]
    assert top.stdout.decode().splitlines() == expect


@pytest.mark.gpu
def test_gpu_mode_topk_never_downloads_the_score_arrays(tmp_path):
    """-k 10 over a 100 000-entry database with a 3-SID query list and solution maps off: the rows
    printed are the sorted head of the full output, and the bytes copied from the GPU are 10 rows per
    query (32 B each) where the full listing copies 400 KB per query."""
    db = sat.synth.make_db(100_000, 8, 32, sort=True, name_format="s%06d")
    db.write_ascii(tmp_path / "db.ascii")
    sids = (db.names[99_999] + "\n" + db.names[50_000] + "\n" + db.names[7] + "\n").encode()
    top = run(CLI, str(tmp_path), ["-r", "64", "-k", "10", "-q", "db.ascii"], stdin_bytes=sids)
    assert top.returncode == 0, top.stderr.decode()[-300:]
    copied = [int(l.split()[1]) for l in top.stderr.decode().splitlines() if l.startswith("copied ")]
    assert copied == [3 * 10 * 32]
    full = run(CLI, str(tmp_path), ["-r", "64", "-q", "db.ascii"], stdin_bytes=sids)
    assert full.returncode == 0
    copied_full = [int(l.split()[1]) for l in full.stderr.decode().splitlines() if l.startswith("copied ")]
    assert copied_full == [3 * 100_000 * 4]
    blocks, cur = [], None
    for line in full.stdout.decode().splitlines():
        if line.startswith("# cudaSaTabsearch"):
            cur = {"head": [line], "rows": []}
            blocks.append(cur)
        elif line.startswith("#"):
            cur["head"].append(line)
        else:
            cur["rows"].append(line)
    expect = []
    for b in blocks:
        rows = sorted(enumerate(b["rows"]), key=lambda t: (-int(t[1].split()[1]), t[0]))[:10]
        expect += b["head"] + [r for _, r in rows]
    assert top.stdout.decode().splitlines() == expect


@pytest.mark.gpu
def test_gpu_mode_output_does_not_depend_on_the_shards(golden_dir):
    """-G 0,0,0: three cost-balanced shards of the database (all on GPU 0 here) give byte for byte the
    stdout of the one-shard run, solution maps included, and the same best-k rows."""
    for name, extra in (("d2phlb1_TTT", []), ("multiquery", []), ("multiquery", ["-k", "5"])):
        one = run(CLI, golden_dir, ["-r", "64", *extra], stdin_path=name + ".input")
        three = run(CLI, golden_dir, ["-r", "64", "-G", "0,0,0", *extra], stdin_path=name + ".input")
        assert one.returncode == 0 and three.returncode == 0, three.stderr.decode()[-300:]
        assert one.stdout == three.stdout
        assert b"to 3 GPU(s)" in three.stderr and b"gather: peer" in three.stderr

"""The host C under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only: GPU ASan is not available on the
pool).  tests/native/satabsearch_asan = csrc/host/{sat_parse, sat_gumbel, sat_shard, sat_host_search, sat_main}.c built
with gcc -fsanitize=address,undefined, the device library's entry points replaced at link time by failing stubs
(tests/native/gpu_stubs.c).  Every run must produce the bytes of the normal build and no sanitizer report."""
import os
import subprocess

import numpy as np
import pytest

import cuda_satabsearch_amd as sat

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "cuda_satabsearch_amd", "csrc", "host")
EXPECTED = os.path.join(ROOT, "tests", "golden", "expected")
ASAN_CLI = os.path.join(ROOT, "tests", "native", "satabsearch_asan")
CLI = os.path.join(ROOT, "cuda_satabsearch_amd", "bin", "satabsearch")


@pytest.fixture(scope="module")
def asan_cli():
    srcs = [os.path.join(HOST, f) for f in ("sat_main.c", "sat_host_search.c", "sat_parse.c", "sat_gumbel.c", "sat_shard.c")]
    srcs.append(os.path.join(ROOT, "tests", "native", "gpu_stubs.c"))
    deps = srcs + [os.path.join(HOST, f) for f in os.listdir(HOST) if f.endswith(".h")]
    if not os.path.exists(ASAN_CLI) or any(os.path.getmtime(d) > os.path.getmtime(ASAN_CLI) for d in deps):
        subprocess.run(["gcc", "-O1", "-g", "-fno-omit-frame-pointer", "-ffp-contract=off", "-fsanitize=address,undefined",
                        "-fno-sanitize-recover=undefined", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"), "-I", HOST,
                        "-o", ASAN_CLI] + srcs + ["-lm", "-lpthread"], check=True)
    return ASAN_CLI


def run(binary, cwd, args, stdin_bytes=b"", env=None):
    e = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:exitcode=99", UBSAN_OPTIONS="print_stacktrace=1")
    e.update(env or {})
    return subprocess.run([binary, *args], input=stdin_bytes, cwd=cwd, capture_output=True, env=e)


def clean(p):
    err = p.stderr.decode(errors="replace")
    assert "ERROR: AddressSanitizer" not in err and "runtime error:" not in err and "LeakSanitizer" not in err, err[-3000:]


@pytest.mark.parametrize("name", ["c1_d1ubia_small", "d2phlb1_TFT", "d2phlb1_TTT", "multiquery", "d1twfa_"])
def test_host_mode_goldens_under_sanitizers(asan_cli, golden_dir, name):
    """the five `-c` goldens of the reference build: byte-identical stdout, no report (r = 128 as in the goldens)"""
    p = run(asan_cli, golden_dir, ["-c", "-r", "128"], open(os.path.join(golden_dir, name + ".input"), "rb").read())
    clean(p)
    assert p.returncode == 0, p.stderr.decode()[-400:]
    assert p.stdout == open(os.path.join(EXPECTED, name + ".r128.out"), "rb").read()


def test_query_list_mode_under_sanitizers(asan_cli, golden_dir):
    sids = open(os.path.join(golden_dir, "qmode_sids.txt"), "rb").read()
    p = run(asan_cli, golden_dir, ["-c", "-r", "16", "-q", "tableauxdistmatrixdb.small.ascii"], sids)
    clean(p)
    assert p.returncode == 0 and p.stdout == open(os.path.join(EXPECTED, "qmode_small.r16.out"), "rb").read()


def test_gpu_mode_fails_cleanly_under_sanitizers(asan_cli, golden_dir):
    p = run(asan_cli, golden_dir, ["-r", "8"], open(os.path.join(golden_dir, "c1_d1ubia_small.input"), "rb").read())
    clean(p)
    assert p.returncode == 1 and b"no usable HIP device" in p.stderr


def test_threaded_reader_and_binary_image_under_sanitizers(asan_cli, tmp_path):
    """10 000 synthetic entries (4.6 MB of text) through the threaded mmap reader (3 and 7 threads, cuts at record
    headers) and through the .satbin image: written, loaded, then TRUNCATED and CORRUPTED images, which the loader
    must refuse (the ASCII file is parsed instead) without touching memory it does not own."""
    db = sat.synth.make_db(10_000, 4, 40, sort=False, seed=21)
    db.write_ascii(tmp_path / "db.ascii")
    q = sat.synth.planted_query(db, 17)
    qs = sat.StructSet.from_dense([len(q[2])], [q[0]], [q[1]], ["QUERY01"])
    qs.write_ascii(tmp_path / "q.body")
    stdin = b"db.ascii\nT T F\n" + open(tmp_path / "q.body", "rb").read()
    ref = run(CLI, str(tmp_path), ["-c", "-r", "1"], stdin)
    assert ref.returncode == 0 and ref.stdout.count(b"\n") == 10_003
    for threads in ("1", "3", "7"):
        p = run(asan_cli, str(tmp_path), ["-c", "-r", "1"], stdin, env={"SAT_PARSE_THREADS": threads})
        clean(p)
        assert p.returncode == 0 and p.stdout == ref.stdout
    first = run(asan_cli, str(tmp_path), ["-c", "-r", "1", "-b"], stdin)          # writes the image
    clean(first)
    image = tmp_path / "db.ascii.satbin"
    assert first.stdout == ref.stdout and image.exists()
    again = run(asan_cli, str(tmp_path), ["-c", "-r", "1", "-b"], stdin)          # loads it
    clean(again)
    assert again.stdout == ref.stdout and b"binary image" in again.stderr
    whole = open(image, "rb").read()
    rng = np.random.default_rng(5)
    for kind in ("truncated", "header", "orders", "offsets"):
        bad = bytearray(whole)
        if kind == "truncated":
            bad = bad[:len(bad) * 2 // 3]
        elif kind == "header":
            bad[8:16] = (2 ** 40).to_bytes(8, "little")           # an absurd count
        else:
            # image = magic[8], {count, cells}[16], orders[4 n], names[9 n], cell offsets[8 n], codes, distances
            lo = 24 if kind == "orders" else 24 + 13 * 10_000
            for k in rng.integers(lo, lo + 4000, size=40):
                bad[int(k)] ^= 0xFF
        open(image, "wb").write(bytes(bad))
        os.utime(image, (os.path.getmtime(tmp_path / "db.ascii") + 10,) * 2)       # newer than the ASCII file
        p = run(asan_cli, str(tmp_path), ["-c", "-r", "1", "-b"], stdin)
        clean(p)
        assert p.returncode == 0 and p.stdout == ref.stdout, (kind, p.stderr.decode()[-300:])

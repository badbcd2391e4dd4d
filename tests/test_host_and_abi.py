"""Host-side logic and the C ABI surface (not gpu: nothing here computes on a device)."""
import ctypes
import os
import re

import numpy as np
import pytest

import cuda_satabsearch_amd as sat
from cuda_satabsearch_amd import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_library_loads_and_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "satabsearch.h")).read()
    declared = set(re.findall(r"\b(sat_[a-z_0-9]+)\s*\(", header))
    assert declared == set(_native.ABI_SYMBOLS)
    lib = ctypes.CDLL(_native.DEVICE_LIB)
    for name in declared:
        assert hasattr(lib, name), name
    assert _native.device_lib().sat_abi_version() == 1


def test_no_device_fails_loudly():
    """Without a HIP device the product refuses to run; it never computes on the CPU."""
    if sat.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(sat.SatError, match="no HIP device"):
        sat.Searcher(0)


def test_reader_fixed_columns_and_codes(golden_dir):
    qs = sat.StructSet.read(os.path.join(golden_dir, "d1ubia_.input"), "query", skip_header_lines=2)
    assert qs.names == ["D1UBIA_"] and list(qs.orders) == [8]
    t, d = qs.dense(0)
    assert list(np.diagonal(t)) == [0, 0, 1, 3, 0, 0, 3, 0]          # e e xa xg e e xg e
    assert t[1, 0] == 0x23 and t[2, 0] == 0x30 and t[3, 0] == 0x01    # OT LE PD
    assert t[0, 1] == t[1, 0]
    assert d[1, 0] == np.float32(4.501) and d[7, 6] == np.float32(15.689)
    assert d[2, 2] == 1.0 and d[3, 3] == 3.0


def test_reader_small_db_shape(golden_dir):
    db = sat.StructSet.read(os.path.join(golden_dir, "tableauxdistmatrixdb.small.ascii"))
    assert len(db) == 586
    assert db.orders.min() == 1 and db.orders.max() == 67
    assert db.names[0] == "d1kcul1" and db.names[4] == "d1nldl1"
    assert int(db.cell_off[-1] + db.orders[-1] * (db.orders[-1] + 1) // 2) == db.tab.size
    # '??' codes (angle computation failed) are nibble 4
    assert (db.tab == 0x44).sum() > 0


def test_reader_seven_column_quirk(golden_dir):
    """Distances >= 100 A are written 7 wide and shift the fixed 7-column parse of the
    rest of that row (parsetableaux.c:288); kept for parity with existing outputs."""
    qs = sat.StructSet.read(os.path.join(golden_dir, "d1twfa_.input"), "query", skip_header_lines=2)
    assert list(qs.orders) == [101]
    lines = open(os.path.join(golden_dir, "d1twfa_.input")).read().splitlines()
    rows = lines[3 + 101:3 + 202]
    t, d = qs.dense(0)
    shifted = 0
    for i, row in enumerate(rows):
        toks = [float(x) for x in row.split()]
        for j in range(i + 1):
            expect = np.float32(float(row[7 * j:].split()[0])) if row[7 * j:].split() else np.float32(0)
            assert d[i, j] == expect
            if np.float32(toks[j]) != expect:
                shifted += 1
    assert shifted > 0


def test_multiquery_reader(golden_dir):
    qs = sat.StructSet.read(os.path.join(golden_dir, "multiquery.input"), "query", skip_header_lines=2)
    assert list(qs.orders) == [8, 13, 101]


def test_gumbel_columns_match_reference_rows():
    # rows of the reference output: d1kcul1 9 (n1=8, n2=12), d1nldl1 11 (n1=8, n2=11)
    assert sat.report.result_lines(["d1kcul1"], [12], [9], 8) == ["d1kcul1  9 0.9 -1.27278 0.943444"]
    assert sat.report.result_lines(["d1nldl1"], [11], [11], 8) == ["d1nldl1  11 1.15789 0.903563 0.161558"]
    assert sat.report.result_lines(["d1ndda_"], [8], [54], 8) == ["d1ndda_  54 6.75 11.7853 1.53059e-07"]


def test_synth_is_chunk_and_shard_invariant():
    whole = sat.synth.make_db(2500, 8, 32)
    part = sat.synth.make_db(700, 8, 32, first_index=900, total=2500)
    ref = whole.subset(np.arange(900, 1600))
    assert np.array_equal(ref.orders, part.orders)
    assert np.array_equal(ref.tab, part.tab) and np.array_equal(ref.dist, part.dist)
    assert (np.diff(whole.orders) >= 0).all()
    assert whole.dist.max() < 100.0


def test_ascii_round_trip(tmp_path):
    db = sat.synth.make_db(40, 3, 20)
    path = tmp_path / "rt.ascii"
    sat.synth.write_ascii(db, path)
    back = sat.StructSet.read(path)
    assert np.array_equal(back.orders, db.orders)
    assert np.array_equal(back.tab, db.tab) and np.array_equal(back.dist, db.dist)
    assert back.names == db.names


def test_fast_distance_cell_equals_strtof_on_its_whole_domain():
    """The mmap reader's fast path turns "ddd.ddd" into float(v / 1000.0); strtof rounds the
    decimal text directly.  Equal for every value the fast path accepts (v < 10^6)."""
    import ctypes
    host = _native.host_lib()
    libc = ctypes.CDLL(None)
    libc.strtof.restype = ctypes.c_float
    libc.strtof.argtypes = [ctypes.c_char_p, ctypes.c_void_p]
    v = np.arange(1_000_000, dtype=np.int64)
    fast = (v.astype(np.float64) / 1000.0).astype(np.float32)          # what distance_at computes
    # numpy parses decimal text correctly rounded, like strtof; spot-check strtof itself below
    text = np.char.add(np.char.add((v // 1000).astype(str), "."), np.char.zfill((v % 1000).astype(str), 3))
    ref = text.astype(np.float32)
    assert np.array_equal(fast, ref)
    for s in [b"  0.000 ", b" 99.999 ", b"123.456 ", b"  4.501  0.000 ", b"  7.5", b"-1.250 ", b"1e2 ", b" nan ", b"", b"   "]:
        a, b = host.sat_distance_cell(s), libc.strtof(s, None)
        assert (a == b) or (np.isnan(a) and np.isnan(b)), s


def test_mmap_reader_equals_stdio_reader(golden_dir, tmp_path):
    for name in ["tableauxdistmatrixdb.small.ascii", "tableauxdistmatrixdb.test2.ascii", "d1qlpa_.ascii"]:
        a = sat.StructSet.read(os.path.join(golden_dir, name))
        b = sat.StructSet.read(os.path.join(golden_dir, name), stdio=True)
        assert a.names == b.names and np.array_equal(a.orders, b.orders)
        assert np.array_equal(a.tab, b.tab) and np.array_equal(a.dist, b.dist)
    # the >= 100 A quirk and an oversized structure, through both readers
    body = open(os.path.join(golden_dir, "d1twfa_.input")).read().split("\n", 2)[2]
    big = sat.synth.make_db(1, 20)
    (tmp_path / "q.ascii").write_text(body)
    a = sat.StructSet.read(tmp_path / "q.ascii")
    b = sat.StructSet.read(tmp_path / "q.ascii", stdio=True)
    assert np.array_equal(a.dist, b.dist) and np.array_equal(a.tab, b.tab)
    lines = ["toobig   112"] + ["e  " * (i + 1) for i in range(112)] + [" 0.000 " * (i + 1) for i in range(112)] + [""]
    sat.synth.write_ascii(big, tmp_path / "tail.ascii")
    (tmp_path / "mixed.ascii").write_text("\n".join(lines) + "\n" + (tmp_path / "tail.ascii").read_text())
    a = sat.StructSet.read(tmp_path / "mixed.ascii")
    b = sat.StructSet.read(tmp_path / "mixed.ascii", stdio=True)
    assert len(a) == len(b) == 1 and a.names == big.names and np.array_equal(a.dist, big.dist)


def test_binary_image_round_trip(tmp_path):
    db = sat.synth.make_db(300, 3, 40)
    db.save_binary(tmp_path / "db.satbin")
    back = sat.StructSet.load_binary(tmp_path / "db.satbin")
    assert back.names == db.names and np.array_equal(back.orders, db.orders)
    assert np.array_equal(back.tab, db.tab) and np.array_equal(back.dist, db.dist) and np.array_equal(back.cell_off, db.cell_off)
    bad = bytearray((tmp_path / "db.satbin").read_bytes())
    bad[40] ^= 0x7F                                         # corrupt an order
    (tmp_path / "bad.satbin").write_bytes(bytes(bad))
    with pytest.raises(OSError):
        sat.StructSet.load_binary(tmp_path / "bad.satbin")


def test_lds_carve_alignment_and_monotonicity():
    """The SA kernel's LDS carve (one function for the kernel and for the launch sizing): the 64-bit
    reduction keys / LSOLN leader key sit on 8-byte boundaries for EVERY shape (a 64-bit LDS atomic on
    a 4-byte aligned address faulted in round 1 when an odd map-word count shifted them), cells and
    query cells on 16, regions do not overlap, and the total grows with the entry order and the query
    order - so a workgroup sized for the launch's largest member holds every member."""
    lib = _native.device_lib()
    out = (ctypes.c_uint32 * 11)()

    def layout(m2w, n1, n1p, n2, chains, threads, qlds, compact):
        lib.sat_debug_lds_layout(m2w, n1, n1p, n2, chains, threads, qlds, compact, out)
        return list(out)

    classes = [(16, range(1, 17)), (32, range(17, 33)), (64, range(33, 65)), (112, range(65, 112))]
    for n1p, n1s in classes:
        for n1 in list(n1s)[::3] + [n1s[-1]]:
            for chains, lpc in ((64, 0), (128, 0), (192, 0), (256, 0), (64, 2), (128, 1)):
                threads = chains << lpc
                for qlds in (0, 1):
                    for compact in (0, 1):
                        prev_total = 0
                        for n2 in list(range(1, 112, 5)) + [32, 33, 111]:
                            m2w = 1 if n2 <= 32 else (2 if n2 <= 64 else 4)
                            code, qdist, qcode, smap, tmask, qtypes, leader, red, red_stride, items, total = layout(
                                m2w, n1, n1p, n2, chains, threads, qlds, compact)
                            assert leader % 8 == 0 and red % 8 == 0 and red_stride % 8 == 0, (n1, n2, chains)
                            assert qdist % 16 == 0 and smap % 4 == 0 and tmask % 4 == 0 and items % 4 == 0
                            assert code <= qdist <= qcode <= smap < tmask < qtypes < leader < red <= items <= total
                            assert tmask - smap >= 4 * ((n1 + 3) // 4) * (chains + 1)
                            assert qtypes - tmask == 16 * m2w and leader - qtypes >= n1p and red - leader == 8
                            waves = (threads + 63) // 64
                            if compact:        # a wave's arg-max key is the head of its own item table
                                assert red == items and red_stride == 256 and total - items == waves * 256
                            else:
                                assert red_stride == 8 and items - red == 16 * 8 and total == items
                            assert red + (waves - 1) * red_stride + 8 <= total
                        # monotone in n2 within a cell layout (m2w | (1 + layout) << 8: 0 = 8-byte cells of the full
                        # matrix, 1 = full matrix in two arrays, 2 = lower triangle in two arrays), and in n1
                        for m2w, cells, orders in ((1, 0, range(1, 33)), (2, 1, range(1, 49)), (2, 2, range(1, 65)), (4, 2, range(1, 112))):
                            totals = [layout(m2w | (1 + cells) << 8, n1, n1p, n2, chains, threads, qlds, compact)[10] for n2 in orders]
                            assert totals == sorted(totals)
                        totals = [layout(1, k, n1p, 20, chains, threads, qlds, compact)[10] for k in n1s]
                        assert totals == sorted(totals)


def test_parallel_reader_equals_sequential(tmp_path, monkeypatch):
    """The mmap reader cuts big files at record headers and parses the pieces on several threads; the
    result must be the sequential reader's, entry for entry - also when a piece does not end at its cut
    (here: a stray one-token line in the middle, at which the reference grammar stops reading), where the
    sequential parse is run again and decides."""
    db = sat.synth.make_db(12_000, 4, 40, sort=False, seed=13)
    path = tmp_path / "db.ascii"
    db.write_ascii(path)
    assert os.path.getsize(path) > 8 << 20
    monkeypatch.setenv("SAT_PARSE_THREADS", "1")
    seq = sat.StructSet.read(path)
    assert len(seq) == len(db) and np.array_equal(seq.tab, db.tab) and np.array_equal(seq.dist, db.dist)
    for threads in ("2", "3", "7", "16"):
        monkeypatch.setenv("SAT_PARSE_THREADS", threads)
        par = sat.StructSet.read(path)
        assert par.names == seq.names and np.array_equal(par.orders, seq.orders)
        assert np.array_equal(par.cell_off, seq.cell_off)
        assert np.array_equal(par.tab, seq.tab) and np.array_equal(par.dist, seq.dist)
    # a stray line after the 7000th record: reading stops there (fscanf("%8s %d") fails)
    text = open(path).read()
    records = text.split("\n\n")
    broken = "\n\n".join(records[:7000]) + "\n\nstray\n\n" + "\n\n".join(records[7000:])
    bad = tmp_path / "stray.ascii"
    open(bad, "w").write(broken)
    monkeypatch.setenv("SAT_PARSE_THREADS", "1")
    seq = sat.StructSet.read(bad)
    assert len(seq) == 7000
    monkeypatch.setenv("SAT_PARSE_THREADS", "8")
    par = sat.StructSet.read(bad)
    assert par.names == seq.names and np.array_equal(par.tab, seq.tab) and np.array_equal(par.dist, seq.dist)


@pytest.mark.parametrize("blank_first_row", [False, True])
def test_parallel_reader_is_not_fooled_by_rows_shaped_like_headers(tmp_path, monkeypatch, blank_first_row):
    """Hand-written databases may hold integer-formatted distance rows: "12 0" has the shape of a record header
    ("name order").  The threaded reader must not cut there: a candidate needs a blank line before it, and if
    even that is met (a record whose first distance row is an empty line) the piece before the false cut ends
    inside a record, is reported incomplete, and the sequential parse - which defines the result - takes over;
    a piece that starts at a false header must never end the process on the "bad code" it then meets."""
    rec = "e  \nPE e  \n" + ("\n" if blank_first_row else "0\n") + "12 0\n\n"
    text = "".join("s%07d    2\n%s" % (k, rec) for k in range(110_000))
    path = tmp_path / "int_rows.ascii"
    open(path, "w").write(text)
    assert os.path.getsize(path) > 3 << 20
    monkeypatch.setenv("SAT_PARSE_THREADS", "1")
    seq = sat.StructSet.read(path)
    assert len(seq) == 110_000 and set(seq.orders.tolist()) == {2}
    assert np.array_equal(seq.dist[:3], np.array([0.0, 12.0, 0.0], np.float32))
    for threads in ("2", "3", "5"):
        monkeypatch.setenv("SAT_PARSE_THREADS", threads)
        par = sat.StructSet.read(path)
        assert par.names == seq.names and np.array_equal(par.orders, seq.orders)
        assert np.array_equal(par.tab, seq.tab) and np.array_equal(par.dist, seq.dist)


def test_ascii_writer_round_trip(tmp_path):
    """sat_set_write_ascii writes the database builder's format (scripts/convdb2.py:214-226): the same
    bytes as the independent Python writer, and the reader gives the structures back exactly - all four
    SSE types, every code the builder can emit plus '??', distances at both ends of %6.3f."""
    db = sat.synth.make_db(300, 1, 60, sort=False, seed=3)
    tab = db.tab.copy()
    off = np.nonzero(tab > 3)[0]
    tab[off[::17]] = 0x44                                   # '??' cells
    dist = db.dist.copy()
    dist[off[::23]] = np.float32(99.999)
    dist[off[::29]] = np.float32(0.001)
    db = sat.StructSet(db.orders, db.names, db.cell_off, tab, dist)
    a, b = tmp_path / "c.ascii", tmp_path / "py.ascii"
    db.write_ascii(a)
    sat.synth.write_ascii(db, b)
    assert open(a, "rb").read() == open(b, "rb").read()
    back = sat.StructSet.read(a)
    assert back.names == db.names and np.array_equal(back.orders, db.orders)
    assert np.array_equal(back.tab, db.tab) and np.array_equal(back.dist, db.dist)
    back2 = sat.StructSet.read(a, stdio=True)
    assert np.array_equal(back2.tab, db.tab) and np.array_equal(back2.dist, db.dist)


def test_readers_drop_records_of_non_positive_order(tmp_path):
    """A record of order 0 (or below) has no rows: both readers drop it with a warning (the reference
    keeps it and then indexes with the order) and the binary image refuses it."""
    db = sat.synth.make_db(3, 5, 5, seed=9)
    good = tmp_path / "good.ascii"
    db.write_ascii(good)
    text = open(good).read()
    bad = tmp_path / "bad.ascii"
    open(bad, "w").write("empty0      0\n\n" + text + "neg1       -3\n\n")
    for stdio in (False, True):
        s = sat.StructSet.read(bad, stdio=stdio)
        assert s.names == db.names and np.array_equal(s.tab, db.tab)

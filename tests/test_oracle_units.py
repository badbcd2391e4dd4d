"""Unit checks of the oracle's building blocks (not gpu)."""
import numpy as np

import oracle_lib


def test_philox_known_answers():
    """Random123 known-answer vectors for philox4x32-10."""
    assert oracle_lib.philox4x32_10([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert oracle_lib.philox4x32_10([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert oracle_lib.philox4x32_10([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_drand48_restatement_matches_libc():
    import ctypes
    libc = ctypes.CDLL(None)
    libc.drand48.restype = ctypes.c_double
    libc.srand48(1234)
    x = oracle_lib.lib().sa_oracle_srand48(1234)
    for _ in range(1000):
        x = (0x5DEECE66D * x + 0xB) & ((1 << 48) - 1)
        assert libc.drand48() == x / 2.0 ** 48


def test_uniform_conversion_range():
    f = oracle_lib.lib().sa_oracle_u32_to_uniform
    assert f(0) == np.float32(2.0 ** -32)
    assert f(0xFFFFFFFF) == 1.0
    assert 0.49 < f(0x7FFFFFFF) <= 0.5


def test_pair_score_table():
    ps = oracle_lib.lib().sa_oracle_pair_score
    assert ps(0x23, 0x23) == 2 and ps(0x23, 0x21) == 1 and ps(0x23, 0x13) == 1 and ps(0x23, 0x10) == -2
    assert ps(0x44, 0x44) == 2 and ps(0x44, 0x04) == 1 and ps(0x44, 0x00) == -2


def test_philox_mode_is_schedule_independent():
    """Scores of an entry depend only on (seed, query, db ordinal): any subset / order of
    entries gives the same per-entry result - the property multi-GPU sharding relies on."""
    import cuda_satabsearch_amd as sat
    db = sat.synth.make_db(40, 5, 20)
    qt, qd, qtypes = sat.synth.planted_query(db, 30)
    full, fmaps, _ = oracle_lib.search(db, qt, qd, qtypes, True, True, 32)
    sub = np.array([30, 3, 17])
    part, pmaps, _ = oracle_lib.search(db, qt, qd, qtypes, True, True, 32, entries=sub)
    assert np.array_equal(part, full[sub]) and np.array_equal(pmaps, fmaps[sub])
    assert full[30] == full.max() and full[30] > 20


def test_sixteen_bit_index_draw_equals_its_integer_form():
    """An SA step's index draws are 16-bit: u = (v + 1) * 2^-16, index = (int)((u - EPS) * n) in
    double, the reference's expression (K.cu:1042, 710).  The GPU kernel evaluates the integer
    ((v + 1) * n - 1) >> 16 instead (sat_sa_kernel.hpp, scaled_index16): equal for every 16-bit v and
    every n the path can see (1..111), checked here exhaustively against the oracle's conversion."""
    lib = oracle_lib.lib()
    import ctypes
    lib.sa_oracle_u16_to_uniform.restype = ctypes.c_float
    lib.sa_oracle_u16_to_uniform.argtypes = [ctypes.c_uint32]
    v = np.arange(65536, dtype=np.int64)
    u = ((v + 1).astype(np.float32) * np.float32(2.0 ** -16))
    for probe in (0, 1, 12345, 65535):
        assert lib.sa_oracle_u16_to_uniform(probe) == u[probe]
    assert u[0] == np.float32(2.0 ** -16) and u[-1] == 1.0
    eps = 1.1e-7
    for n in range(1, 112):
        ref = ((u.astype(np.float64) - eps) * n).astype(np.int64)          # C's (int) truncation: values are >= 0
        integer = ((v + 1) * n - 1) >> 16
        assert np.array_equal(ref, integer), n
        assert integer.max() == n - 1 and integer.min() == 0


def test_initial_map_draw_threshold_in_integers():
    """thinit matches a query SSE when u < 0.5 (K.cu:625); for u = 2^-32 + float(v) * 2^-32 that is
    v < 0x7FFFFFC0 (float(v) rounds to 2^31 from there on): the form the kernel tests."""
    f = oracle_lib.lib().sa_oracle_u32_to_uniform
    for v in list(range(0x7FFFFF00, 0x80000100)) + [0, 1, 0x3FFFFFFF, 0x7FFFFFBF, 0x7FFFFFC0, 0xFFFFFFFF]:
        assert (f(v) < 0.5) == (v < 0x7FFFFFC0), hex(v)

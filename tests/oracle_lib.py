"""ctypes access to oracle/liboracle.so - the CPU oracle (test infrastructure).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "liboracle.so")
MAXDIM = 111
RNG_DRAND48, RNG_PHILOX = 0, 1


class _Rng(C.Structure):
    _fields_ = [("mode", C.c_int), ("lcg", C.c_uint64), ("seed", C.c_uint64), ("query_ordinal", C.c_uint32)]


class _Query(C.Structure):
    _fields_ = [("n", C.c_int), ("pitch", C.c_int), ("tab", C.c_void_p), ("dmat", C.c_void_p),
                ("ssetypes", C.c_void_p)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            import subprocess
            subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all"], check=True)
        l = C.CDLL(LIB)
        l.sa_oracle_srand48.restype = C.c_uint64
        l.sa_oracle_srand48.argtypes = [C.c_long]
        l.sa_oracle_u32_to_uniform.restype = C.c_float
        l.sa_oracle_u32_to_uniform.argtypes = [C.c_uint32]
        l.sa_oracle_pair_score.argtypes = [C.c_uint8, C.c_uint8]
        l.sa_oracle_search.restype = None
        _lib = l
    return _lib


def philox4x32_10(counter, key):
    c = (C.c_uint32 * 4)(*counter)
    k = (C.c_uint32 * 2)(*key)
    out = (C.c_uint32 * 4)()
    lib().sa_oracle_philox4x32_10(c, k, out)
    return list(out)


def search(db, qtab, qdmat, qtypes, lorder=True, lsoln=False, maxstart=128, mode=RNG_PHILOX, seed=1234,
           query_ordinal=0, db_ordinal=None, lcg=None, entries=None):
    """Run the oracle over `db` (a StructSet) or the subset `entries` of it.

    Returns (scores int32[n], ssemaps int32[n,111] or None, lcg_state)."""
    l = lib()
    idx = np.arange(len(db)) if entries is None else np.asarray(entries)
    n = len(idx)
    pitch = int(db.orders[idx].max())
    tabs = np.zeros((n, pitch, pitch), np.uint8)
    dmats = np.zeros((n, pitch, pitch), np.float32)
    for k, s in enumerate(idx):
        t, d = db.dense(int(s), pitch)
        tabs[k], dmats[k] = t, d
    orders = np.ascontiguousarray(db.orders[idx], dtype=np.int32)
    ordinal = np.ascontiguousarray(idx if db_ordinal is None else np.asarray(db_ordinal)[idx], dtype=np.int64)
    qtab = np.ascontiguousarray(qtab, np.uint8)
    qdmat = np.ascontiguousarray(qdmat, np.float32)
    qtypes = np.ascontiguousarray(qtypes, np.uint8)
    q = _Query(qtab.shape[0], qtab.shape[1], qtab.ctypes.data, qdmat.ctypes.data, qtypes.ctypes.data)
    rng = _Rng(mode, l.sa_oracle_srand48(1234) if lcg is None else lcg, seed, query_ordinal)
    scores = np.empty(n, np.int32)
    ssemaps = np.full((n, MAXDIM), -1, np.int32)
    l.sa_oracle_search(C.byref(q), n, orders.ctypes.data_as(C.c_void_p), ordinal.ctypes.data_as(C.c_void_p),
                       tabs.ctypes.data_as(C.c_void_p), dmats.ctypes.data_as(C.c_void_p), pitch,
                       int(bool(lorder)), int(bool(lsoln)), int(maxstart), C.byref(rng),
                       scores.ctypes.data_as(C.c_void_p), ssemaps.ctypes.data_as(C.c_void_p))
    return scores, (ssemaps if lsoln else None), rng.lcg

"""GPU search: thin object wrapper over the C ABI (include/satabsearch.h).

Argument names and meaning follow the reference kernel contract
(nvcc_src_current/cudaSaTabsearch_kernel.cu:756-802): lorder, lsoln, maxstart,
scores per db entry, ssemap[entry][query SSE] = matched db SSE or -1.
"""
import ctypes as C

import numpy as np

from . import _native
from ._native import SatError
from .structures import StructSet

MAXDIM = _native.MAXDIM
DEFAULT_MAXSTART = 128   # saparams.h:40
DEFAULT_SEED = 1234      # cudaSaTabsearch.cu:263, :871


def device_count():
    return int(_native.device_lib().sat_device_count())


class Searcher:
    """One HIP device, one resident database shard, one current query."""

    def __init__(self, device=0, seed=DEFAULT_SEED):
        self._lib = _native.device_lib()
        self._ctx = self._lib.sat_ctx_create(int(device), int(seed))
        if not self._ctx:
            raise SatError(self._lib.sat_last_error().decode())
        self.device = int(device)
        self.n_entries = 0
        self.n1 = 0

    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.sat_ctx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise SatError(f"[{rc}] {self._lib.sat_last_error().decode()}")

    # ---- database -----------------------------------------------------------
    def upload(self, db: StructSet, db_ordinal=None):
        """Upload a (shard of a) database.  db_ordinal[e] = position of entry e in the
        whole database's file order; it keys the random streams so that sharding does
        not change results.  Default: 0..N-1."""
        n = len(db)
        ordinal = None
        if db_ordinal is not None:
            ordinal = np.ascontiguousarray(db_ordinal, dtype=np.int64)
            if ordinal.shape[0] != n:
                raise ValueError("db_ordinal length mismatch")
        self._check(self._lib.sat_db_upload_packed(
            self._ctx, n, db.orders.ctypes.data, db.cell_off.ctypes.data, db.tab.ctypes.data,
            db.dist.ctypes.data, ordinal.ctypes.data if ordinal is not None else None))
        self.n_entries = n
        self._orders = db.orders.copy()

    def upload_search(self, db: StructSet, lorder=True, lsoln=False, maxstart=DEFAULT_MAXSTART, db_ordinal=None):
        """upload() and the first search of the query (batch) set before, overlapped: each piece of the
        shard is searched while the next one is copied (sat_db_upload_search).  Collect with results()."""
        n = len(db)
        ordinal = None
        if db_ordinal is not None:
            ordinal = np.ascontiguousarray(db_ordinal, dtype=np.int64)
            if ordinal.shape[0] != n:
                raise ValueError("db_ordinal length mismatch")
        self.n_entries = 0
        self._check(self._lib.sat_db_upload_search(
            self._ctx, n, db.orders.ctypes.data, db.cell_off.ctypes.data, db.tab.ctypes.data,
            db.dist.ctypes.data, ordinal.ctypes.data if ordinal is not None else None,
            int(bool(lorder)), int(bool(lsoln)), int(maxstart)))
        self.n_entries = n
        self._orders = db.orders.copy()

    def upload_dense(self, orders, tabs, dmats, pitch, db_ordinal=None):
        orders = np.ascontiguousarray(orders, dtype=np.int32)
        tabs = np.ascontiguousarray(tabs, dtype=np.uint8)
        dmats = np.ascontiguousarray(dmats, dtype=np.float32)
        ordinal = None if db_ordinal is None else np.ascontiguousarray(db_ordinal, dtype=np.int64)
        self._check(self._lib.sat_db_upload_dense(
            self._ctx, orders.shape[0], orders.ctypes.data, tabs.ctypes.data, dmats.ctypes.data, int(pitch),
            ordinal.ctypes.data if ordinal is not None else None))
        self.n_entries = int(orders.shape[0])
        self._orders = orders.copy()

    # ---- query --------------------------------------------------------------
    def set_query(self, qtab, qdmat, qssetypes=None, query_ordinal=0):
        """qtab / qdmat: dense [n1, P] matrices (P >= n1); SSE types default to the
        tableau diagonal (cudaSaTabsearch.cu:410-412)."""
        qtab = np.ascontiguousarray(qtab, dtype=np.uint8)
        qdmat = np.ascontiguousarray(qdmat, dtype=np.float32)
        n1, pitch = qtab.shape[0], qtab.shape[1]
        if qdmat.shape != qtab.shape:
            raise ValueError("qtab and qdmat shapes differ")
        if qssetypes is None:
            qssetypes = np.ascontiguousarray(np.diagonal(qtab)[:n1])
        qssetypes = np.ascontiguousarray(qssetypes, dtype=np.uint8)
        self._check(self._lib.sat_query_set(self._ctx, n1, qtab.ctypes.data, qdmat.ctypes.data, pitch,
                                            qssetypes.ctypes.data, int(query_ordinal)))
        self.n1 = n1
        self.n_queries = 1
        self._batch = False

    def set_queries(self, queries, first_query_ordinal=0):
        """Set a batch of queries scored together by one search(): `queries` is a list of
        (qtab[n1, n1+], qdmat, qssetypes) triples.  search() then returns scores[nq, N]
        (and ssemaps[nq, N, 111]); query q draws from the streams of ordinal
        first_query_ordinal + q."""
        nq = len(queries)
        pitch = max(int(np.asarray(q[0]).shape[0]) for q in queries)
        n1s = np.empty(nq, np.int32)
        tabs = np.zeros((nq, pitch, pitch), np.uint8)
        dmats = np.zeros((nq, pitch, pitch), np.float32)
        types = np.zeros((nq, pitch), np.uint8)
        for k, (t, d, ty) in enumerate(queries):
            t = np.asarray(t, np.uint8)
            d = np.asarray(d, np.float32)
            n1 = t.shape[0]
            n1s[k] = n1
            tabs[k, :n1, :n1] = t[:, :n1]
            dmats[k, :n1, :n1] = d[:, :n1]
            types[k, :n1] = np.asarray(ty, np.uint8)[:n1] if ty is not None else np.diagonal(t)[:n1]
        self._check(self._lib.sat_queries_set(self._ctx, nq, n1s.ctypes.data, tabs.ctypes.data, dmats.ctypes.data,
                                              pitch, types.ctypes.data, int(first_query_ordinal)))
        self.n1 = int(n1s[0])
        self.n_queries = nq
        self._batch = True

    def set_query_from(self, queries: StructSet, s, query_ordinal=None):
        t, d = queries.dense(s)
        self.set_query(t, d, queries.ssetypes(s), s if query_ordinal is None else query_ordinal)

    # ---- search -------------------------------------------------------------
    def search(self, lorder=True, lsoln=False, maxstart=DEFAULT_MAXSTART):
        """Returns (scores int32[N], ssemaps int32[N, 111] or None, kernel_ms)."""
        nq = getattr(self, "n_queries", 1)
        scores = np.empty((nq, self.n_entries), np.int32)
        ssemaps = np.full((nq, self.n_entries, MAXDIM), -1, np.int32) if lsoln else None
        ms = C.c_double(0.0)
        self._check(self._lib.sat_search(self._ctx, int(bool(lorder)), int(bool(lsoln)), int(maxstart),
                                         scores.ctypes.data, ssemaps.ctypes.data if lsoln else None,
                                         C.byref(ms)))
        if not getattr(self, "_batch", False):
            return scores[0], (ssemaps[0] if lsoln else None), ms.value
        return scores, ssemaps, ms.value

    def use_stream(self, stream_handle):
        """Queue all further work on the caller's HIP stream (0 / None = default stream),
        e.g. torch.cuda.current_stream().cuda_stream."""
        self._check(self._lib.sat_use_stream(self._ctx, C.c_void_p(int(stream_handle or 0))))

    def use_own_stream(self):
        self._check(self._lib.sat_use_own_stream(self._ctx))

    def search_async(self, lorder=True, lsoln=False, maxstart=DEFAULT_MAXSTART):
        """Queue the search on the context's current stream (no sync, no copy); results
        stay in device memory."""
        self._check(self._lib.sat_search_async(self._ctx, int(bool(lorder)), int(bool(lsoln)), int(maxstart)))

    def results(self, lsoln=False):
        """Wait for a queued search_async and fetch (scores, ssemaps or None)."""
        nq = getattr(self, "n_queries", 1)
        scores = np.empty((nq, self.n_entries), np.int32)
        ssemaps = np.full((nq, self.n_entries, MAXDIM), -1, np.int32) if lsoln else None
        self._check(self._lib.sat_results(self._ctx, int(bool(lsoln)), scores.ctypes.data,
                                          ssemaps.ctypes.data if lsoln else None))
        if not getattr(self, "_batch", False):
            return scores[0], (ssemaps[0] if lsoln else None)
        return scores, ssemaps

    def topk(self, k, query=0):
        """(entry_index int32[k'], scores int32[k']) of the best k hits of the last search,
        sorted on the device by descending score, ties in database order."""
        idx = np.empty(k, np.int32)
        sc = np.empty(k, np.int32)
        n = self._lib.sat_topk(self._ctx, int(query), int(k), idx.ctypes.data, sc.ctypes.data)
        if n < 0:
            self._check(n)
        return idx[:n], sc[:n]

    def topk_hits(self, k, lsoln=False):
        """Best-k rows of every query of the last search, ranked on the device with their statistics:
        a structured array [nq, k'] with fields entry, score, norm2, zscore, pvalue (and the rows'
        solution maps int32[nq, k', 111] when lsoln)."""
        nq = getattr(self, "n_queries", 1)
        k = min(int(k), self.n_entries)
        dt = np.dtype([("entry", np.int32), ("score", np.int32), ("norm2", np.float64), ("zscore", np.float64),
                       ("pvalue", np.float64)], align=True)
        assert dt.itemsize == C.sizeof(_native.Hit)
        hits = np.zeros((nq, k), dt)
        maps = np.full((nq, k, MAXDIM), -1, np.int32) if lsoln else None
        n = self._lib.sat_topk_hits(self._ctx, k, hits.ctypes.data, maps.ctypes.data if lsoln else None)
        if n < 0:
            self._check(n)
        return (hits, maps) if lsoln else hits

    def d2h_bytes(self):
        """Bytes this context's result calls have copied device -> host so far."""
        return int(self._lib.sat_stat_d2h_bytes(self._ctx))

    def last_launch_info(self):
        """Kernel instantiations and launch geometry of the last search."""
        return self._lib.sat_last_launch_info(self._ctx).decode()

    def sync(self):
        self._check(self._lib.sat_sync(self._ctx))

    def search_timed(self, lorder=True, lsoln=False, maxstart=DEFAULT_MAXSTART, repeats=1):
        """HIP-event time of `repeats` back-to-back searches on the launch stream (ms)."""
        total, kern = C.c_double(0.0), C.c_double(0.0)
        self._check(self._lib.sat_search_timed(self._ctx, int(bool(lorder)), int(bool(lsoln)), int(maxstart),
                                               int(repeats), C.byref(total), C.byref(kern)))
        return total.value, kern.value

    def device_scores_ptr(self):
        return self._lib.sat_device_scores(self._ctx)

    def device_scores_tensor(self):
        """int32 torch tensor aliasing the context's device score buffer (for RCCL gathers)."""
        import torch

        class _Holder:
            pass

        h = _Holder()
        h.__cuda_array_interface__ = {
            "shape": (getattr(self, "n_queries", 1) * self.n_entries,), "typestr": "<i4", "data": (int(self.device_scores_ptr()), False),
            "version": 3, "strides": None,
        }
        return torch.as_tensor(h, device=f"cuda:{self.device}")


class MultiSearcher:
    """One search over several GPUs from one host thread (sat_multi_* of the C ABI): cost-balanced
    contiguous shards, the queries on every GPU, one gather of the shard rows to device 0."""

    def __init__(self, ndev=0, devices=None, seed=DEFAULT_SEED):
        self._lib = _native.device_lib()
        dev = None if devices is None else np.ascontiguousarray(devices, dtype=np.int32)
        self._m = self._lib.sat_multi_create(int(ndev), dev.ctypes.data if dev is not None else None, int(seed))
        if not self._m:
            raise SatError(self._lib.sat_last_error().decode())
        self.ndev = int(self._lib.sat_multi_device_count(self._m))
        self.n_entries = 0
        self.n_queries = 0

    def close(self):
        if getattr(self, "_m", None):
            self._lib.sat_multi_destroy(self._m)
            self._m = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc < 0:
            raise SatError(f"[{rc}] {self._lib.sat_last_error().decode()}")

    @property
    def gather_kind(self):
        return self._lib.sat_multi_gather_kind(self._m).decode()

    def upload(self, db: StructSet):
        self._check(self._lib.sat_multi_db_upload_packed(self._m, len(db), db.orders.ctypes.data, db.cell_off.ctypes.data,
                                                         db.tab.ctypes.data, db.dist.ctypes.data))
        self.n_entries = len(db)

    def shards(self):
        begin = np.zeros(self.ndev + 1, np.int32)
        self._check(self._lib.sat_multi_shards(self._m, begin.ctypes.data))
        return begin

    def set_queries(self, queries, first_query_ordinal=0):
        nq = len(queries)
        pitch = max(int(np.asarray(q[0]).shape[0]) for q in queries)
        n1s = np.empty(nq, np.int32)
        tabs = np.zeros((nq, pitch, pitch), np.uint8)
        dmats = np.zeros((nq, pitch, pitch), np.float32)
        types = np.zeros((nq, pitch), np.uint8)
        for k, (t, d, ty) in enumerate(queries):
            t = np.asarray(t, np.uint8)
            d = np.asarray(d, np.float32)
            n1 = t.shape[0]
            n1s[k] = n1
            tabs[k, :n1, :n1] = t[:, :n1]
            dmats[k, :n1, :n1] = d[:, :n1]
            types[k, :n1] = np.asarray(ty, np.uint8)[:n1] if ty is not None else np.diagonal(t)[:n1]
        self._check(self._lib.sat_multi_queries_set(self._m, nq, n1s.ctypes.data, tabs.ctypes.data, dmats.ctypes.data,
                                                    pitch, types.ctypes.data, int(first_query_ordinal)))
        self.n_queries = nq

    def search(self, lorder=True, lsoln=False, maxstart=DEFAULT_MAXSTART):
        """Returns (scores int32[nq, N], ssemaps int32[nq, N, 111] or None, wall_ms), database order."""
        scores = np.empty((self.n_queries, self.n_entries), np.int32)
        ssemaps = np.full((self.n_queries, self.n_entries, MAXDIM), -1, np.int32) if lsoln else None
        ms = C.c_double(0.0)
        self._check(self._lib.sat_multi_search(self._m, int(bool(lorder)), int(bool(lsoln)), int(maxstart), scores.ctypes.data,
                                               ssemaps.ctypes.data if lsoln else None, C.byref(ms)))
        return scores, ssemaps, ms.value

    def search_topk(self, k, lorder=True, lsoln=False, maxstart=DEFAULT_MAXSTART):
        k = min(int(k), self.n_entries)
        dt = np.dtype([("entry", np.int32), ("score", np.int32), ("norm2", np.float64), ("zscore", np.float64),
                       ("pvalue", np.float64)], align=True)
        hits = np.zeros((self.n_queries, k), dt)
        maps = np.full((self.n_queries, k, MAXDIM), -1, np.int32) if lsoln else None
        ms = C.c_double(0.0)
        self._check(self._lib.sat_multi_search_topk(self._m, int(bool(lorder)), int(bool(lsoln)), int(maxstart), k,
                                                    hits.ctypes.data, maps.ctypes.data if lsoln else None, C.byref(ms)))
        return hits, maps, ms.value

    def d2h_bytes(self):
        return int(self._lib.sat_multi_stat_d2h_bytes(self._m))

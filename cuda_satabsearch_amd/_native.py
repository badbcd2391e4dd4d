"""ctypes bindings of the two in-tree shared libraries.

libsatabsearch.so is the product: HIP kernels behind the C ABI of
include/satabsearch.h.  There is no Python or CPU stand-in for it: if the library
is missing, or no HIP device is usable, the calls raise.
"""
import ctypes as C
import os

PKG = os.path.dirname(os.path.abspath(__file__))
# SAT_DEVICE_LIB lets kernel experiments (scripts/exp/) load an alternative build of the same ABI
DEVICE_LIB = os.environ.get("SAT_DEVICE_LIB") or os.path.join(PKG, "libsatabsearch.so")
HOST_LIB = os.path.join(PKG, "libsathost.so")

MAXDIM = 111
LABELSIZE = 8

# every symbol include/satabsearch.h declares
ABI_SYMBOLS = (
    "sat_last_error", "sat_abi_version", "sat_device_count", "sat_ctx_create", "sat_ctx_destroy",
    "sat_db_upload_packed", "sat_db_upload_search", "sat_db_upload_dense", "sat_db_size", "sat_query_set", "sat_search",
    "sat_search_async", "sat_device_scores", "sat_device_ssemaps", "sat_query_order", "sat_sync",
    "sat_search_timed", "sat_use_stream", "sat_use_own_stream", "sat_results", "sat_queries_set", "sat_query_count", "sat_topk",
    "sat_topk_hits", "sat_stat_d2h_bytes", "sat_debug_lds_layout", "sat_last_launch_info",
    "sat_multi_create", "sat_multi_destroy", "sat_multi_device_count", "sat_multi_gather_kind", "sat_multi_db_upload_packed",
    "sat_multi_shards", "sat_multi_queries_set", "sat_multi_search", "sat_multi_search_topk", "sat_multi_stat_d2h_bytes",
)


class SatError(RuntimeError):
    pass


class Hit(C.Structure):
    """struct sat_hit of include/satabsearch.h"""
    _fields_ = [("entry", C.c_int32), ("score", C.c_int32), ("norm2", C.c_double), ("zscore", C.c_double),
                ("pvalue", C.c_double)]


class StructSetC(C.Structure):
    """struct sat_struct_set of csrc/host/sat_parse.h"""
    _fields_ = [
        ("count", C.c_int), ("capacity", C.c_int),
        ("cells", C.c_int64), ("cells_cap", C.c_int64),
        ("order", C.POINTER(C.c_int)), ("name", C.POINTER(C.c_char)),
        ("cell_off", C.POINTER(C.c_int64)), ("tab", C.POINTER(C.c_uint8)),
        ("dist", C.POINTER(C.c_float)), ("skipped", C.c_int),
    ]


_device = None
_host = None


def device_lib():
    """Load libsatabsearch.so (raises SatError when it has not been built)."""
    global _device
    if _device is None:
        if not os.path.exists(DEVICE_LIB):
            raise SatError(f"{DEVICE_LIB} is missing: run `python -m cuda_satabsearch_amd.build` "
                           "(the GPU search has no fallback implementation)")
        lib = C.CDLL(DEVICE_LIB)
        lib.sat_last_error.restype = C.c_char_p
        lib.sat_abi_version.restype = C.c_int
        lib.sat_device_count.restype = C.c_int
        lib.sat_ctx_create.restype = C.c_void_p
        lib.sat_ctx_create.argtypes = [C.c_int, C.c_uint64]
        lib.sat_ctx_destroy.argtypes = [C.c_void_p]
        lib.sat_ctx_destroy.restype = None
        lib.sat_db_upload_packed.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p]
        lib.sat_db_upload_search.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        lib.sat_db_upload_dense.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_int, C.c_void_p]
        lib.sat_db_size.argtypes = [C.c_void_p]
        lib.sat_query_set.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                      C.c_uint32]
        lib.sat_search.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                   C.POINTER(C.c_double)]
        lib.sat_search_async.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        lib.sat_use_stream.argtypes = [C.c_void_p, C.c_void_p]
        lib.sat_use_own_stream.argtypes = [C.c_void_p]
        lib.sat_results.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        lib.sat_queries_set.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                        C.c_void_p, C.c_uint32]
        lib.sat_query_count.argtypes = [C.c_void_p]
        lib.sat_topk.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        lib.sat_topk_hits.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        lib.sat_stat_d2h_bytes.argtypes = [C.c_void_p]
        lib.sat_stat_d2h_bytes.restype = C.c_uint64
        lib.sat_last_launch_info.argtypes = [C.c_void_p]
        lib.sat_last_launch_info.restype = C.c_char_p
        lib.sat_debug_lds_layout.argtypes = [C.c_int] * 8 + [C.c_void_p]
        lib.sat_debug_lds_layout.restype = None
        lib.sat_multi_create.restype = C.c_void_p
        lib.sat_multi_create.argtypes = [C.c_int, C.c_void_p, C.c_uint64]
        lib.sat_multi_destroy.argtypes = [C.c_void_p]
        lib.sat_multi_destroy.restype = None
        lib.sat_multi_device_count.argtypes = [C.c_void_p]
        lib.sat_multi_gather_kind.argtypes = [C.c_void_p]
        lib.sat_multi_gather_kind.restype = C.c_char_p
        lib.sat_multi_db_upload_packed.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.sat_multi_shards.argtypes = [C.c_void_p, C.c_void_p]
        lib.sat_multi_queries_set.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_uint32]
        lib.sat_multi_search.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
        lib.sat_multi_search_topk.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                              C.POINTER(C.c_double)]
        lib.sat_multi_stat_d2h_bytes.argtypes = [C.c_void_p]
        lib.sat_multi_stat_d2h_bytes.restype = C.c_uint64
        lib.sat_device_scores.argtypes = [C.c_void_p]
        lib.sat_device_scores.restype = C.c_void_p
        lib.sat_device_ssemaps.argtypes = [C.c_void_p]
        lib.sat_device_ssemaps.restype = C.c_void_p
        lib.sat_query_order.argtypes = [C.c_void_p]
        lib.sat_sync.argtypes = [C.c_void_p]
        lib.sat_search_timed.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.POINTER(C.c_double), C.POINTER(C.c_double)]
        _device = lib
    return _device


def host_lib():
    """Load libsathost.so (ASCII reader + Gumbel statistics, plain C)."""
    global _host
    if _host is None:
        if not os.path.exists(HOST_LIB):
            raise SatError(f"{HOST_LIB} is missing: run `python -m cuda_satabsearch_amd.build`")
        lib = C.CDLL(HOST_LIB)
        lib.sat_set_init.argtypes = [C.POINTER(StructSetC)]
        lib.sat_set_init.restype = None
        lib.sat_set_free.argtypes = [C.POINTER(StructSetC)]
        lib.sat_set_free.restype = None
        lib.sat_read_structures.argtypes = [C.c_void_p, C.POINTER(StructSetC), C.c_char_p]
        lib.sat_read_structures.restype = C.c_int
        lib.sat_read_structures_file.argtypes = [C.c_char_p, C.POINTER(StructSetC), C.c_char_p]
        lib.sat_read_structures_file.restype = C.c_int
        lib.sat_set_save_binary.argtypes = [C.POINTER(StructSetC), C.c_char_p]
        lib.sat_set_load_binary.argtypes = [C.c_char_p, C.POINTER(StructSetC)]
        lib.sat_set_write_ascii.argtypes = [C.POINTER(StructSetC), C.c_char_p]
        lib.sat_distance_cell.argtypes = [C.c_char_p]
        lib.sat_distance_cell.restype = C.c_float
        lib.sat_norm2.argtypes = [C.c_int, C.c_int, C.c_int]
        lib.sat_norm2.restype = C.c_double
        lib.sat_z_gumbel_trunc.argtypes = [C.c_double]
        lib.sat_z_gumbel_trunc.restype = C.c_double
        lib.sat_pv_gumbel.argtypes = [C.c_double]
        lib.sat_pv_gumbel.restype = C.c_double
        lib.sat_entry_cost.argtypes = [C.c_int]
        lib.sat_entry_cost.restype = C.c_double
        lib.sat_shard_cuts.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        _host = lib
    return _host


_libc = C.CDLL(None)
_libc.fopen.restype = C.c_void_p
_libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
_libc.fclose.argtypes = [C.c_void_p]
_libc.fgets.restype = C.c_void_p
_libc.fgets.argtypes = [C.c_char_p, C.c_int, C.c_void_p]


def libc():
    return _libc

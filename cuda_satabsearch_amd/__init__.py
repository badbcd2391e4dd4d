"""cuda_satabsearch_amd - MI355X (gfx950) implementation of the cudaSaTabsearch
simulated-annealing tableau search hot path.

The compute path is libsatabsearch.so (hand-written HIP kernel behind the C ABI of
include/satabsearch.h).  This package is the host-side mirror used by tests and
bench.py: structure sets (packed database / queries), the Searcher wrapper, result
formatting and the seeded synthetic database generator.
"""
from ._native import SatError, ABI_SYMBOLS  # noqa: F401
from .structures import StructSet, MAXDIM, MAXDIM_SMALL  # noqa: F401
from .search import Searcher, MultiSearcher, device_count, DEFAULT_MAXSTART, DEFAULT_SEED  # noqa: F401
from . import report, sharding, synth, workloads  # noqa: F401

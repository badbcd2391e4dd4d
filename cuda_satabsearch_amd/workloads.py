"""The named workloads of BASELINE.json `configs` at their stated sizes,
shared by tests/test_gpu_configs.py, scripts/run_config.py and the profiling scripts.

configs[2]  d2phlb1.input + the three multiquery.input queries (orders 19, 8, 13, 101) against a
            synthetic, size-sorted 100 000-entry database of 8..32-SSE structures, r = 4096 - the
            reference's own heavy workload (old/nvcc_src_cuda5/fermi_qlist_*.e1462446), inline and
            as a `-q` SID list
configs[3]  32-SSE synthetic query against 1 000 000 synthetic 32-SSE entries, r = 128 (2.7 GB packed:
            fits one MI355X; bench.py runs the 125 000-entry-per-GPU shards of the same database)
configs[4]  d1twfa_.input (101 SSEs) against 100 000 entries of the large-structure generator C5
            (uniform [8, 96] + 1 % in [97, 111], size sorted), LSOLN = T, r = 128

The query files are the reference's example inputs, held as data under tests/golden/inputs.
"""
import os

import numpy as np

from . import synth
from .structures import StructSet

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN_INPUTS = os.path.join(ROOT, "tests", "golden", "inputs")


def load_queries(filename, golden_dir=GOLDEN_INPUTS):
    """Every query of a reference-format input file: [(name, tab, dist, ssetypes)]."""
    qs = StructSet.read(os.path.join(golden_dir, filename), "query", skip_header_lines=2)
    out = []
    for k in range(len(qs)):
        t, d = qs.dense(k)
        out.append((qs.names[k], t, d, qs.ssetypes(k)))
    return out


def config2_db(n=100_000):
    # 7-character names: `-q` cuts SIDs to 7 characters (cudaSaTabsearch.cu:633-635, 657)
    return synth.make_db(n, 8, 32, sort=True, name_format="s%06d")


def config2_queries(golden_dir=GOLDEN_INPUTS):
    return load_queries("d2phlb1.input", golden_dir) + load_queries("multiquery.input", golden_dir)


def config3_db(n=1_000_000):
    return synth.make_db(n, 32)


def config3_query():
    return synth.make_query(32)


def config4_db(n=100_000):
    return synth.make_db(n, orders=synth.orders_c5(n))


def config4_query(golden_dir=GOLDEN_INPUTS):
    return load_queries("d1twfa_.input", golden_dir)[0]


def mixed_db(n=100_000):
    """C3-style database used for the mixed-size throughput figure: orders uniform on [8, 32],
    size sorted (as `convdb2.py -s` writes real databases)."""
    return synth.make_db(n, 8, 32, sort=True)


def sample_entries(n, k=48, seed=0):
    return np.sort(np.random.default_rng(seed).choice(n, k, replace=False))

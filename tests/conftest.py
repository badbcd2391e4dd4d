import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_built():
    """Host library, device library (cross-compiles without a GPU) and the oracle."""
    from cuda_satabsearch_amd import build
    build.build_host()
    build.build_device()
    build.build_cli()
    build.build_test_native()
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all"], check=True)
    yield


@pytest.fixture(scope="session")
def golden_dir(tmp_path_factory):
    """tests/golden/inputs unpacked into a temp dir (query files name their database by
    relative path, and the 586-entry database is stored gzipped)."""
    import gzip
    import shutil
    src = os.path.join(ROOT, "tests", "golden", "inputs")
    dst = tmp_path_factory.mktemp("golden_inputs")
    for f in os.listdir(src):
        if f.endswith(".gz"):
            with gzip.open(os.path.join(src, f), "rb") as fi, open(os.path.join(dst, f[:-3]), "wb") as fo:
                shutil.copyfileobj(fi, fo)
        else:
            shutil.copy(os.path.join(src, f), os.path.join(dst, f))
    return str(dst)

"""Build the native pieces of cuda_satabsearch_amd in-tree.

    python -m cuda_satabsearch_amd.build [--oracle] [--ref]

* libsatabsearch.so   HIP kernels + C ABI (include/satabsearch.h), hipcc --offload-arch=gfx950
* libsathost.so       host-only C: ASCII reader + Gumbel statistics (csrc/host/)
* bin/satabsearch     the command line (csrc/host/sat_main.c), links both

`--oracle` additionally runs oracle/Makefile (test infrastructure, never linked into
the product), `--ref` also its `ref` target when /root/reference is mounted.
hipcc cross-compiles gfx950 code objects without a GPU present.
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
HOST = os.path.join(CSRC, "host")
INC = os.path.join(ROOT, "include")

HIPCC = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
CC = os.environ.get("CC") or "gcc"


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def kernel_source_hash():
    """sha256 over the sources of the SA kernel and its dispatch: profiles/bench_traffic.json carries it, so that
    bench.py only reports committed counter figures that were measured on the kernel it is running."""
    import hashlib
    h = hashlib.sha256()
    for f in ("sat_sa_kernel.hpp", "sat_capi.hip", "sat_ctx.hpp"):
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def build_host(force=False):
    out = os.path.join(PKG, "libsathost.so")
    srcs = [os.path.join(HOST, f) for f in ("sat_parse.c", "sat_gumbel.c", "sat_shard.c")]
    deps = srcs + [os.path.join(HOST, f) for f in ("sat_parse.h", "sat_gumbel.h", "sat_shard.h")]
    if force or _stale(out, deps):
        _run([CC, "-O2", "-fPIC", "-shared", "-Wall", "-Wextra", "-I", HOST, "-o", out] + srcs + ["-lm", "-lpthread"])
    return out


def _host_objects(force=False):
    """plain-C host pieces linked into the device library (and its diagnostic twin): the Gumbel statistics (the
    context tabulates them with the host libm for the device-side best-k rows) and the shard cost model"""
    host_c = [os.path.join(HOST, "sat_gumbel.c"), os.path.join(HOST, "sat_shard.c")]
    host_o = [os.path.join(PKG, "sat_gumbel.o"), os.path.join(PKG, "sat_shard.o")]
    for c, o in zip(host_c, host_o):
        if force or _stale(o, [c, os.path.join(HOST, "sat_gumbel.h"), os.path.join(HOST, "sat_shard.h")]):
            _run([CC, "-O2", "-fPIC", "-ffp-contract=off", "-Wall", "-Wextra", "-I", HOST, "-c", "-o", o, c])
    return host_o


def build_device(force=False):
    out = os.path.join(PKG, "libsatabsearch.so")
    srcs = [os.path.join(CSRC, "sat_capi.hip"), os.path.join(CSRC, "sat_topk.hip"), os.path.join(CSRC, "sat_multi.hip")]
    host_o = _host_objects(force)
    deps = srcs + host_o + [os.path.join(CSRC, "sat_sa_kernel.hpp"), os.path.join(CSRC, "sat_ctx.hpp"),
                            os.path.join(INC, "satabsearch.h")]
    if force or _stale(out, deps):
        # -Wl,: hipcc would compile a bare .o as HIP source.  librccl is NOT linked: sat_multi.hip loads it on demand
        _run([HIPCC, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared",
              "-I", INC, "-I", CSRC, "-o", out] + srcs + ["-Wl," + o for o in host_o] + ["-lm", "-ldl"])
    return out


def build_cli(force=False):
    src = os.path.join(HOST, "sat_main.c")
    if not os.path.exists(src):
        return None
    bindir = os.path.join(PKG, "bin")
    os.makedirs(bindir, exist_ok=True)
    out = os.path.join(bindir, "satabsearch")
    host_search = os.path.join(HOST, "sat_host_search.c")
    deps = [src, host_search, os.path.join(HOST, "sat_host_search.h"),
            os.path.join(PKG, "libsatabsearch.so"), os.path.join(PKG, "libsathost.so")]
    if force or _stale(out, deps):
        # -O3 without fast-math or fma contraction: the host mode must round like the reference's
        _run([CC, "-O3", "-ffp-contract=off", "-Wall", "-Wextra", "-I", INC, "-I", HOST, "-o", out, src, host_search,
              "-L", PKG, "-lsatabsearch", "-lsathost", "-lm", "-Wl,-rpath,$ORIGIN/.."])
    return out


def build_test_native(force=False):
    """tests/native/*.hip: GPU-side TEST helpers (e.g. the rocRAND device API beside the kernel's own
    Philox block).  Test infrastructure like oracle/: never loaded by the product."""
    tdir = os.path.join(ROOT, "tests", "native")
    src = os.path.join(tdir, "rocrand_check.hip")
    if not os.path.exists(src):
        return None
    out = os.path.join(tdir, "librocrand_check.so")
    if force or _stale(out, [src, os.path.join(CSRC, "sat_sa_kernel.hpp")]):
        _run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I", INC, "-I", CSRC, "-o", out, src])
    # the device library once more with the reference's TESTING assertion compiled in (diag/sat_diag.hpp,
    # -DSAT_DIAG_SELFCHECK: every proposed move's score against a full recomputation) - loaded only by
    # tests/test_gpu_parity.py::test_every_move_passes_the_references_self_check, through SAT_DEVICE_LIB
    out3 = os.path.join(tdir, "libsat_selfcheck.so")
    dsrcs = [os.path.join(CSRC, f) for f in ("sat_capi.hip", "sat_topk.hip", "sat_multi.hip")]
    ddeps = dsrcs + [os.path.join(CSRC, "sat_sa_kernel.hpp"), os.path.join(CSRC, "sat_ctx.hpp"),
                     os.path.join(CSRC, "diag", "sat_diag.hpp"), os.path.join(INC, "satabsearch.h")]
    if force or _stale(out3, ddeps):
        _run([HIPCC, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared", "-DSAT_DIAG",
              "-DSAT_DIAG_SELFCHECK", "-I", INC, "-I", CSRC, "-o", out3] + dsrcs +
             ["-Wl," + os.path.join(PKG, "sat_gumbel.o"), "-Wl," + os.path.join(PKG, "sat_shard.o"), "-lm", "-ldl"])
    src2, out2 = os.path.join(tdir, "lds_residency.hip"), os.path.join(tdir, "liblds_residency.so")
    if os.path.exists(src2) and (force or _stale(out2, [src2])):
        _run([HIPCC, "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", out2, src2])
    return out


def build_oracle(ref=False):
    odir = os.path.join(ROOT, "oracle")
    _run(["make", "-s", "-C", odir, "all"])
    if ref and os.path.isdir("/root/reference/nvcc_src_current"):
        _run(["make", "-s", "-C", odir, "ref"])


def build_all(force=False, oracle=False, ref=False):
    build_host(force)
    _host_objects(force)
    if oracle:
        # the device library and its diagnostic twin (tests/native/libsat_selfcheck.so: the same sources with the
        # reference's per-move assertion compiled in) take ~100 s of hipcc each: side by side
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(2) as ex:
            dev = ex.submit(build_device, False if not force else True)
            nat = ex.submit(build_test_native, force)
            dev.result()
            nat.result()
        build_cli(force)
        build_oracle(ref)
    else:
        build_device(force)
        build_cli(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, oracle="--oracle" in sys.argv or "--ref" in sys.argv,
              ref="--ref" in sys.argv)

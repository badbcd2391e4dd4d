#!/bin/bash
# scripts/exp/variant_lib.sh NAME [hipcc flags...] - build a variant of the device library as
# cuda_satabsearch_amd/libsat_NAME.so (git-ignored; travels to the GPU box) for A/B runs:
#   scripts/exp/variant_lib.sh phase -DSAT_DIAG -DSAT_DIAG_PHASE      per-phase wave-cycle table on stderr after each search
#   scripts/exp/variant_lib.sh p1 -DSAT_DIAG -DSAT_DIAG_PERTURB=1       +40 full-rate VALU per SA step (2: SALU, 3: LDS, 4: s_nop, 5: half-rate VALU)
# then on the GPU box:  SAT_DEVICE_LIB=cuda_satabsearch_amd/libsat_NAME.so python scripts/quick_bench.py 125000 32 32 32 5
set -e
repo=$(cd "$(dirname "$0")/../.." && pwd)
name=$1; shift
# same inputs as cuda_satabsearch_amd/build.py build_device (the host-C objects come from a normal build)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fPIC -shared "$@" \
  -I $repo/include -I $repo/cuda_satabsearch_amd/csrc -o $repo/cuda_satabsearch_amd/libsat_$name.so \
  $repo/cuda_satabsearch_amd/csrc/sat_capi.hip $repo/cuda_satabsearch_amd/csrc/sat_topk.hip $repo/cuda_satabsearch_amd/csrc/sat_multi.hip \
  -Wl,$repo/cuda_satabsearch_amd/sat_gumbel.o -Wl,$repo/cuda_satabsearch_amd/sat_shard.o -lm -ldl
echo "built cuda_satabsearch_amd/libsat_$name.so"

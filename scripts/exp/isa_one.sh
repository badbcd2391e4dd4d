#!/bin/bash
# scripts/exp/isa_one.sh OUT.s [N1P M2W QLDS OPT WPL CELLS] - device assembly of one kernel instantiation (default <32, 1, false, 1, 4, 0>)
set -e
repo=$(cd "$(dirname "$0")/../.." && pwd)
out=${1:-/tmp/sat_kernel.s}
tmp=$(mktemp /tmp/sat_one_XXXX.hip)
echo "#include \"sat_sa_kernel.hpp\"" > $tmp
echo "template __global__ void sat_sa_kernel<${2:-32}, ${3:-1}, ${4:-false}, ${5:-1}, ${6:-4}, ${7:-0}>(const SatKernelArgs);" >> $tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -I$repo/cuda_satabsearch_amd/csrc -I$repo/include --offload-device-only -S -o $out $tmp 2>&1 | grep -v "warning\|^$" || true
rm -f $tmp
grep "\.vgpr_count\|\.sgpr_spill_count\|\.vgpr_spill_count\|\.sgpr_count" $out
echo "VALU lines: $(grep -c '^\s*v_' $out)"

#!/bin/bash
# Build timing-only ablation variants of libsatabsearch.so (never shipped): scripts/exp/lib_<name>.so
set -e
cd "$(dirname "$0")/../.."
for v in "$@"; do
  name=${v%%:*}; defs=${v#*:}
  flags=""
  for d in ${defs//,/ }; do flags="$flags -D$d"; done
  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fPIC -shared $flags -Iinclude -Icuda_satabsearch_amd/csrc cuda_satabsearch_amd/csrc/sat_topk.hip \
    -o scripts/exp/lib_$name.so cuda_satabsearch_amd/csrc/sat_capi.hip &
done
wait
ls -la scripts/exp/*.so

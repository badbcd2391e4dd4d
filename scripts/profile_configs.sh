#!/bin/bash
# scripts/profile_configs.sh TAG [workloads...] - rocprofv3 --kernel-trace --stats of the named
# workloads of scripts/run_config.py (default: c2 c4 c3_1m mixed lorderf n96), one run each.
# Output: gpurun_out/prof_TAG/<workload>/ ; copy the *_kernel_stats.csv into profiles/.
set -uo pipefail
tag=${1:-r02}; shift || true
wl=${@:-c2 c4 c3_1m mixed lorderf n96}
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for w in $wl; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/$w" -- python3 $repo/scripts/run_config.py $w 2 > "$out/$w.log" 2>&1
  echo "$w rc=$?"; tail -1 "$out/$w.log"
  f=$(find "$out/$w" -name "*_kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$out/${w}_kernel_stats.csv"
  # the per-dispatch trace is large: keep only the stats table
  find "$out/$w" -name "*_kernel_trace.csv" -delete
done

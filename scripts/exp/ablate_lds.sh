#!/bin/bash
# scripts/exp/ablate_lds.sh - which LDS access site makes the bank conflicts of the bench kernel?
# Builds diagnostic variants of the device library in which ONE access site is issued twice (-DSAT_DIAG_DUP =
# 1: the db-cell gathers of the pair evaluation, 2: the chain-map words of the rounds, 3: the item
# accumulator atomics, 4: the own-map byte read of the proposal; volatile duplicates, results
# unchanged) and, on the GPU box, collects SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE / SQ_INSTS_LDS and
# the kernel time of the bench workload for each: a site's share is the growth over the base build.
# (Removing or redirecting an access instead changes the SA trajectories and with them the work.)
#   build (container):  scripts/exp/ablate_lds.sh build
#   run (GPU box):      scripts/exp/ablate_lds.sh run   -> gpurun_out/ablate_lds.txt
set -uo pipefail
repo=$(cd "$(dirname "$0")/../.." && pwd)
if [ "${1:-}" = build ]; then
  for k in 1 2 3 4; do bash $repo/scripts/exp/variant_lib.sh abl$k -DSAT_DIAG -DSAT_DIAG_DUP=$k & done; wait
  exit 0
fi
out=$repo/gpurun_out/ablate_lds.txt; : > $out
cd /tmp && export TMPDIR=/tmp
for v in base abl1 abl2 abl3 abl4; do
  lib=$repo/cuda_satabsearch_amd/libsat_$v.so; [ $v = base ] && lib=$repo/cuda_satabsearch_amd/libsatabsearch.so
  export SAT_DEVICE_LIB=$lib
  d=$repo/gpurun_out/abl_$v; rm -rf $d
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $d -- python3 $repo/scripts/quick_bench.py 125000 32 32 32 3 > $d.log 2>&1
  python3 - "$v" "$d" >> $out <<'PY'
import csv, glob, sys, collections
v, d = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for p in glob.glob(d + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(p)):
        if "sat_sa_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(x) / len(x) for k, x in agg.items()}
print(f"{v:5s} conflict {m.get('SQ_LDS_BANK_CONFLICT',0):.4g}  idx_active {m.get('SQ_LDS_IDX_ACTIVE',0):.4g}  "
      f"ratio {m.get('SQ_LDS_BANK_CONFLICT',0)/max(m.get('SQ_LDS_IDX_ACTIVE',1),1):.3f}  insts_lds {m.get('SQ_INSTS_LDS',0):.4g}")
PY
  grep "scorings/s" $d.log | sed "s/^/      /" >> $out
done
cat $out

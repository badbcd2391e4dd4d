#!/bin/bash
# scripts/profile_config_pmc.sh TAG WORKLOAD - rocprofv3 PMC passes (one counter set per run, no trace
# domains) of one workload of scripts/run_config.py.  Output: gpurun_out/pmc_TAG_WORKLOAD/summary.txt
set -uo pipefail
tag=${1:-r02}; w=${2:-c4}
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/pmc_${tag}_$w
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d "$out/p$i" -- python3 $repo/scripts/run_config.py $w 2 > "$out/p$i.log" 2>&1
  echo "pass $i rc=$?"
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob(out + "/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(p)):
        if "sat_sa_" in r["Kernel_Name"]:
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
lines = []
for k, cs in agg.items():
    lines.append(k)
    tot = {c: sum(v) for c, v in cs.items()}
    n = {c: len(v) for c, v in cs.items()}
    for c in sorted(tot):
        lines.append(f"  {c:24s} total {tot[c]:.5g} over {n[c]} dispatches")
    if "SQ_INSTS_VALU" in tot and "GRBM_GUI_ACTIVE" in tot:
        cyc = tot["GRBM_GUI_ACTIVE"] / 8
        lines.append(f"  derived: VALU wave-instr per SIMD-cycle {tot['SQ_INSTS_VALU']/1024/cyc:.3f}; cycles per VALU instr per SIMD {cyc*1024/tot['SQ_INSTS_VALU']:.2f}")
        if "SQ_WAVE_CYCLES" in tot: lines.append(f"  derived: resident waves per SIMD {tot['SQ_WAVE_CYCLES']*4/cyc/1024:.2f}")
        if "SQ_LDS_IDX_ACTIVE" in tot: lines.append(f"  derived: LDS busy {tot['SQ_LDS_IDX_ACTIVE']/256/cyc:.3f}, conflicts/active {tot['SQ_LDS_BANK_CONFLICT']/tot['SQ_LDS_IDX_ACTIVE']:.3f}")
        if "TCP_TCC_READ_REQ_sum" in tot: lines.append(f"  derived: L1->L2 read requests per CU-cycle {tot['TCP_TCC_READ_REQ_sum']/256/cyc:.3f}; L2 hit rate {tot.get('TCC_HIT_sum',0)/max(tot.get('TCC_HIT_sum',0)+tot.get('TCC_MISS_sum',0),1):.3f}")
open(out + "/summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY

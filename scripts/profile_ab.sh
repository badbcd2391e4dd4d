#!/bin/bash
# scripts/profile_ab.sh TAG [LIB] - three PMC passes of the bench workload with the device library LIB
# (default: the in-tree one).  Output: gpurun_out/ab_TAG/
set -uo pipefail
tag=${1:-x}
repo=${GRAFT_REPO_ROOT:-/root/repo}
[ -n "${2:-}" ] && export SAT_DEVICE_LIB=$repo/$2
out=$repo/gpurun_out/ab_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 $repo/bench.py $ARGS > "$out/trace.log" 2>&1
echo "trace rc=$?"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM_RD \
  --output-format csv -d "$out/pmc1" -- python3 $repo/bench.py $ARGS > "$out/pmc1.log" 2>&1
echo "pmc1 rc=$?"
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_IFETCH SQ_INST_CYCLES_SALU \
  --output-format csv -d "$out/pmc2" -- python3 $repo/bench.py $ARGS > "$out/pmc2.log" 2>&1
echo "pmc2 rc=$?"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_THREAD_CYCLES_VALU SQ_LDS_ADDR_CONFLICT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY \
  --output-format csv -d "$out/pmc3" -- python3 $repo/bench.py $ARGS > "$out/pmc3.log" 2>&1
echo "pmc3 rc=$?"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_IOPS \
  --output-format csv -d "$out/pmc4" -- python3 $repo/bench.py $ARGS > "$out/pmc4.log" 2>&1
echo "pmc4 rc=$?"

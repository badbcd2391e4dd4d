#!/bin/bash
# scripts/profile_bench.sh TAG - rocprofv3 runs of the bench workload on the GPU box.
# Kernel trace + stats in one run, PMC counters in runs of their own (gpurun refuses
# --pmc combined with the trace domains).  Output: gpurun_out/prof_TAG/
set -uo pipefail
tag=${1:-r1}
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-upload-probe --no-regimes"
# (the trace run takes 30 timed steps, so that the first launch of the process does not weigh on the average)
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 $repo/bench.py --steps 30 --warmup 2 --no-cpu-baseline --no-upload-probe --no-regimes > "$out/trace.log" 2>&1
echo "trace rc=$?"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD \
  --output-format csv -d "$out/pmc1" -- python3 $repo/bench.py $ARGS > "$out/pmc1.log" 2>&1
echo "pmc1 rc=$?"
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  --output-format csv -d "$out/pmc2" -- python3 $repo/bench.py $ARGS > "$out/pmc2.log" 2>&1
echo "pmc2 rc=$?"
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d "$out/pmc3" -- python3 $repo/bench.py $ARGS > "$out/pmc3.log" 2>&1
echo "pmc3 rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc4" -- python3 $repo/bench.py $ARGS > "$out/pmc4.log" 2>&1
echo "pmc4 rc=$?"
find "$out" -name "*.csv" | head -40
du -sh "$out"

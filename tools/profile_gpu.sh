#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + PMC passes of the bench workload.
# usage: tools/profile_gpu.sh <tag> [records] [steps]
# Output: gpurun_out/prof_<tag>/{kt,pmc1,pmc2,...}/  (copy the summaries you want judged into profiles/)
set -u
TAG=${1:-r01}
REC=${2:-20000000}
STEPS=${3:-3}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --records $REC --steps $STEPS --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- $BENCH > $OUT/kt.log 2>&1
echo "kernel-trace rc=$?"
i=0
for SET in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
  "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH SQ_WAVES" \
  "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/pmc$i -o pmc -- $BENCH > $OUT/pmc$i.log 2>&1
  echo "pmc$i [$SET] rc=$?"
done
python3 $ROOT/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
tail -40 $OUT/summary.txt

#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + PMC passes (each counter set in its own
# run, never combined with a runtime/sys trace) of the bench workload.
# usage: tools/profile_gpu.sh <tag> [records] [steps] [extra bench args...]
# Output: gpurun_out/prof_<tag>/{kt*,pmcN/}, summary.txt, kernel_stats.csv, traffic.json
#         (copy the summaries you want judged into profiles/)
set -u
TAG=${1:-r02}
REC=${2:-100000000}
STEPS=${3:-5}
shift 3 2>/dev/null || true
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--records $REC --steps $STEPS --warmup 1 --no-cpu-baseline --no-other-configs $*"
echo "python3 bench.py $ARGS" > $OUT/command.txt
# the kernel trace runs the bench's own default step counts (20 timed + 3 warm-up launches), so that its
# average is the steady-state figure bench.py reports from hipEvents; the PMC passes use fewer steps
KT_ARGS="--records $REC --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs $*"
echo "python3 bench.py $KT_ARGS   (kernel trace)" >> $OUT/command.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 $ROOT/bench.py $KT_ARGS > $OUT/kt.log 2>&1
echo "kernel-trace rc=$?"
i=0
for SET in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
  "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/pmc$i -o pmc -- python3 $ROOT/bench.py $ARGS > $OUT/pmc$i.log 2>&1
  echo "pmc$i [$SET] rc=$?"
done
python3 $ROOT/tools/summarize_prof.py $OUT "$ARGS" > $OUT/summary.txt 2>&1
tail -50 $OUT/summary.txt

#!/bin/bash
# Runs on the GPU box: `merkurio tag` BAM -> BAM with the diagnostic build of the library that logs every device-buffer growth
# (python -m merkurio_amd.build --tag alloclog --flags "-DMK_ALLOC_LOG=1"): which growths cost what, hipFree against hipMalloc.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
BIN=$ROOT/merkurio_amd/lib/merkurio
[ -f /tmp/e2e_in.bam ] || python3 $ROOT/tools/e2e_tag.py 8000000 > /dev/null 2>&1
for W in 240 1024; do
  echo "== --window-mb $W"
  LD_PRELOAD=$ROOT/merkurio_amd/lib/libmerkurio_hip_alloclog.so MERKURIO_TIMING=1 $BIN tag -f /tmp/e2e_kmers.txt -i /tmp/e2e_in.bam -o /tmp/alloc_out.bam --window-mb $W 2>&1 | grep "alloc\]\|windows on the device"
done

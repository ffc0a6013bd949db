#!/bin/bash
# Runs on the GPU box: `merkurio tag` BAM -> BAM (records resident on the device) on the BAM of tools/e2e_tag.py, by window size.
# usage: tools/tag_window_sweep.sh [records, default 8000000]   (expects /tmp/e2e_in.bam and /tmp/e2e_kmers.txt, or makes them)
N=${1:-8000000}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
BIN=$ROOT/merkurio_amd/lib/merkurio
[ -f /tmp/e2e_in.bam ] || python3 $ROOT/tools/e2e_tag.py $N > /dev/null 2>&1
for W in 128 256 512 1024 4096; do
  for REP in 1 2; do
    T0=$(date +%s.%N)
    MERKURIO_TIMING=1 $BIN tag -f /tmp/e2e_kmers.txt -i /tmp/e2e_in.bam -o /tmp/sweep_out.bam --window-mb $W 2> /tmp/sweep_err.txt
    T1=$(date +%s.%N)
    echo "--window-mb $W: $(python3 -c "print('%.2f s wall' % ($T1 - $T0))"); $(grep 'windows on the device' /tmp/sweep_err.txt | sed 's/\[timing\] //' | tr '\n' ' ')"
  done
done

#!/bin/bash
# round 3, batch 2: full GPU suite on the new ordering path, hits-mode bench lines with ordering inside the
# step, per-kernel split of the ordering (rocprofv3 kernel trace)
set -e
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/r03_pytest_b2.txt
echo "# C4 shape (20 M x 150 bp, 10 k 31-mers, hit tuples, ordered)" | tee gpurun_out/r03_hits_mode.txt
timeout -k 10 300 python bench.py --records 20000000 --mode hits --no-cpu-baseline --steps 10 2>&1 | tail -1 | tee -a gpurun_out/r03_hits_mode.txt
echo "# every read hits (100 M x 150 bp), tuples ordered" | tee -a gpurun_out/r03_hits_mode.txt
timeout -k 10 300 python bench.py --mode hits --plant-every 1 --no-cpu-baseline --steps 5 2>&1 | tail -1 | tee -a gpurun_out/r03_hits_mode.txt
echo "# 10 % of the reads hit" | tee -a gpurun_out/r03_hits_mode.txt
timeout -k 10 300 python bench.py --mode hits --plant-every 10 --no-cpu-baseline --steps 5 2>&1 | tail -1 | tee -a gpurun_out/r03_hits_mode.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_order -o order -- python3 $R/tools/order_hits_bench.py 100000000 > $R/gpurun_out/r03_order_prof.log 2>&1
cd $R
f=$(ls gpurun_out/prof_order/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && head -20 "$f" | tee gpurun_out/r03_order_kernel_stats.csv

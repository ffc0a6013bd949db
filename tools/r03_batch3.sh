#!/bin/bash
# round 3, batch 3: ordering tests after the re-binning fix; fixed-length batches (no offsets lookup) against the
# offsets path over the hit-rate sweep; the headline line
set -e
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_order.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r03_pytest_b3.txt
echo "# fixed record length (record of an occurrence computed)" | tee gpurun_out/r03_hitrate_sweep.txt
bash tools/hitrate_sweep.sh 2>&1 | tee -a gpurun_out/r03_hitrate_sweep.txt
echo "# offsets array (record looked up): the r02 path" | tee -a gpurun_out/r03_hitrate_sweep.txt
bash tools/hitrate_sweep.sh --with-offsets 2>&1 | tee -a gpurun_out/r03_hitrate_sweep.txt
timeout -k 10 300 python bench.py 2>&1 | tail -1 | tee gpurun_out/r03_bench_default_b3.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_order -o order -- python3 $R/tools/order_hits_bench.py 100000000 > $R/gpurun_out/r03_order_prof.log 2>&1
cd $R
f=$(find gpurun_out/prof_order -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -20 "$f" | tee gpurun_out/r03_order_kernel_stats.csv

#!/bin/bash
# the judged set of a round in one call (round 3: profiles/r04_*), on one box and one build -- full GPU suite, default bench line (headline +
# other_configs), rocprofv3 kernel trace + PMC passes of the headline and of the config-5 shard, ordering timings and
# their per-kernel split, hits-mode lines, hit-rate sweep, dense driver-loop timing
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -q -m gpu 2>&1 | tail -4 | tee gpurun_out/r04_pytest_final.txt
timeout -k 10 600 python bench.py 2> gpurun_out/r04_bench_default.err | tail -1 > gpurun_out/r04_bench_default.json
bash tools/profile_gpu.sh r04 100000000 5 > gpurun_out/r04_profile.log 2>&1
bash tools/profile_gpu.sh r04_c5 12500000 5 --read-len 250 --patterns 500000 --k 21 > gpurun_out/r04_profile_c5.log 2>&1
rm -f gpurun_out/r04_order_hits.txt
for n in 1000000 4000000 100000000; do
  timeout -k 10 300 python tools/order_hits_bench.py $n 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04_order_hits.txt
done
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_order -o order -- python3 $R/tools/order_hits_bench.py 100000000 > $R/gpurun_out/r04_order_prof.log 2>&1 )
f=$(find gpurun_out/prof_order -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -16 "$f" > gpurun_out/r04_order_kernel_stats.csv
{ echo "# C4 shape (20 M x 150 bp, 10 k 31-mers, hit tuples, ordered)"; timeout -k 10 300 python bench.py --records 20000000 --mode hits --no-cpu-baseline --steps 10 2>/dev/null | tail -1
  echo "# every read hits (100 M x 150 bp), tuples ordered"; timeout -k 10 300 python bench.py --mode hits --plant-every 1 --no-cpu-baseline --steps 5 2>/dev/null | tail -1
  echo "# 10 % of the reads hit"; timeout -k 10 300 python bench.py --mode hits --plant-every 10 --no-cpu-baseline --steps 5 2>/dev/null | tail -1; } > gpurun_out/r04_hits_mode.txt
{ echo "# fixed record length (record of an occurrence computed)"; bash tools/hitrate_sweep.sh; echo "# offsets array (record looked up): the r02 path"; bash tools/hitrate_sweep.sh --with-offsets; } > gpurun_out/r04_hitrate_sweep.txt 2>&1
timeout -k 10 300 python tools/tag_records_bench.py 4000000 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_tag_records.txt
tail -3 gpurun_out/r04_pytest_final.txt; python -c "
import json; j=json.load(open('gpurun_out/r04_bench_default.json')); print(j['value'], j['roofline']['frac'], [ (o['kernel_ms'], o['frac']) for o in j['other_configs']])"

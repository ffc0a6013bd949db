#!/bin/bash
# round 3, batch 4: full GPU suite on the new flavours / fixed-length path; hit-rate sweep with the 16-byte
# compare twin of the sparse kernel; default bench line
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/r03_pytest_b4.txt
echo "# flavours: sparse < 2 % <= cmp16 < 12 % <= plain; fixed record length" | tee gpurun_out/r03_hitrate_sweep_b4.txt
bash tools/hitrate_sweep.sh 2>&1 | tee -a gpurun_out/r03_hitrate_sweep_b4.txt
for pe in 50 20 10 5; do for hint in 0 50; do
echo -n "plant_every=$pe density-hint=$hint any: " | tee -a gpurun_out/r03_hitrate_sweep_b4.txt
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 5 --warmup 2 --plant-every $pe --density-hint $hint 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('step', j['ms_per_step'], 'kernel', r['kernel_ms_avg'], j['config']['kernel'])" | tee -a gpurun_out/r03_hitrate_sweep_b4.txt
done; done
timeout -k 10 300 python bench.py 2>&1 | tail -1 | tee gpurun_out/r03_bench_default_b4.json

#!/bin/bash
# 16-byte level-3 loads in the plain-load (dense) variants only: parity, then the crossover between
# the sparse (nt loads, 8-byte compare) and dense (plain loads, 16-byte compare) variants by hit rate
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/b23_tests.log 2>&1 || { tail -30 gpurun_out/b23_tests.log; exit 1; }
tail -2 gpurun_out/b23_tests.log
for pe in 0 30 20 10 7 5 3 1; do for mode in any hits; do for hint in 0 1000; do for r in 1 2; do
  echo -n "pe=$pe mode=$mode hint=$hint: "
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 8 --warmup 2 --plant-every $pe --mode $mode --density-hint $hint 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['roofline']['kernel_ms_avg'], j['config']['kernel'])" || exit 1
done; done; done; done 2>&1 | tee gpurun_out/b23_crossover.txt

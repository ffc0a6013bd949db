#!/bin/bash
# crossover between the sparse-hit kernels (nt loads, record lists) and the dense ones (plain loads, 16-byte compare,
# direct flag stores) by hit rate, with bench.py --density-hint forcing either flavour
cd ${GRAFT_REPO_ROOT:-/root/repo}
for pe in 10 7 5 4 3 2 1; do for mode in any hits; do for hint in 0 1000; do for r in 1 2; do
  echo -n "pe=$pe mode=$mode hint=$hint: "
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 8 --warmup 2 --plant-every $pe --mode $mode --density-hint $hint 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['roofline']['kernel_ms_avg'], j['config']['kernel'])" || exit 1
done; done; done; done 2>&1 | tee gpurun_out/r02_crossover2.txt

#!/bin/bash
# hit-dense flavour: probe / drain earlier so that level 3's text re-read still finds its line in L2
# (build first: python -m merkurio_amd.build --tag d32 --flags "-DMK_ISSUE_AT_DENSE=32 -DMK_DRAIN_AT_DENSE=32", d16, d8)
run() { echo -n "$1: "; shift; timeout -k 10 300 python bench.py --no-cpu-baseline --no-other-configs --steps 5 --warmup 2 "$@" 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('kernel', j['roofline']['kernel_ms_avg'], 'ms', j['config']['kernel'], 'hits', j['summary']['hits']//5)"; }
L=$PWD/merkurio_amd/lib
for r in 1 2; do for pe in 1 3; do for v in default d16 d16f d8f; do
  if [ $v = default ]; then unset MERKURIO_LIB_PATH; else export MERKURIO_LIB_PATH=$L/libmerkurio_hip_$v.so; fi
  run "plant_every=$pe $v any " --plant-every $pe --density-hint 1000
done; done; done
for v in default d16f; do
  if [ $v = default ]; then unset MERKURIO_LIB_PATH; else export MERKURIO_LIB_PATH=$L/libmerkurio_hip_$v.so; fi
  run "plant_every=1 $v hits" --plant-every 1 --density-hint 1000 --mode hits --no-order
done

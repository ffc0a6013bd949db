#!/bin/bash
# where does a true hit's time go?  headline workload (1 % of reads hit), same box, ablation builds
run() { echo -n "$1: "; shift; timeout -k 10 300 python bench.py --no-cpu-baseline --steps 10 "$@" 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms_avg'])"; }
L=merkurio_amd/lib
for r in 1 2; do
unset MERKURIO_LIB_PATH
run "full, 1% reads hit            "
run "full, 1%, no counters         " --no-counters
run "full, no planted hits         " --plant-every 0
export MERKURIO_LIB_PATH=$L/libmerkurio_hip_abl32.so
run "level 3 without stores (32)   "
export MERKURIO_LIB_PATH=$L/libmerkurio_hip_abl16.so
run "level 3 dropped (16)          "
export MERKURIO_LIB_PATH=$L/libmerkurio_hip_abl64.so
run "q-gram hits not queued (64)   "
done

#!/bin/bash
# where does the time of a hit-dense batch go?  headline batch with every read / 10 % of the reads hitting, the
# kernel flavour pinned with --density-hint, ablation builds (tools/build_ablations.sh 1 8 16 64 first):
#   64 = q-gram hits not queued (no level 3 at all), 16 = level 3 dropped (hits queued, never resolved),
#   8 = filter positives queued but never probed (no level 2 / 3), 1 = every filter positive dropped
run() { echo -n "$1: "; shift; timeout -k 10 300 python bench.py --no-cpu-baseline --no-other-configs --steps 5 --warmup 2 "$@" 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('kernel', j['roofline']['kernel_ms_avg'], 'ms', j['config']['kernel'], 'hits', j['summary']['hits']//5, 'candidates', j['summary']['filter_candidates']//5)"; }
L=$PWD/merkurio_amd/lib
for spec in "1 1000" "10 100"; do set -- $spec
  for a in full 64 16 8 1; do
    if [ $a = full ]; then unset MERKURIO_LIB_PATH; else export MERKURIO_LIB_PATH=$L/libmerkurio_hip_abl$a.so; fi
    run "plant_every=$1 hint=$2 $a" --plant-every $1 --density-hint $2
  done
done
unset MERKURIO_LIB_PATH
run "no hits, sparse kernel" --plant-every 0 --density-hint 0
run "no hits, plain kernel " --plant-every 0 --density-hint 1000

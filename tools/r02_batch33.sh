#!/bin/bash
# A/B only (see r02_batch32.sh), any-mode and hits-mode, on whatever box comes up
cd ${GRAFT_REPO_ROOT:-/root/repo}
L=$PWD/merkurio_amd/lib
for r in 1 2; do for v in prev new; do
  if [ $v = new ]; then unset MERKURIO_LIB_PATH; else export MERKURIO_LIB_PATH=$L/libmerkurio_hip_$v.so; fi
  for pe in 0 100 10; do for mode in any hits; do
  echo -n "$v pe=$pe $mode: "
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --plant-every $pe --mode $mode --density-hint 0 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['roofline']['kernel_ms_avg'], j['roofline']['kernel_ms_min'], j['config']['kernel'], j['summary']['records_hit'])" || exit 1
  done; done
done; done

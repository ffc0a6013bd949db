#!/bin/bash
# per-wave lists of flagged records in the sparse-hit kernels: full GPU suite, then A/B against the previous build
cd ${GRAFT_REPO_ROOT:-/root/repo}
( time timeout -k 10 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/r02_gputest32.log 2>&1 || { tail -30 gpurun_out/r02_gputest32.log; exit 1; }
tail -4 gpurun_out/r02_gputest32.log
L=$PWD/merkurio_amd/lib
for r in 1 2; do for v in prev new; do
  if [ $v = new ]; then unset MERKURIO_LIB_PATH; else export MERKURIO_LIB_PATH=$L/libmerkurio_hip_$v.so; fi
  for pe in 0 1000 100 20; do for mode in any hits; do
  echo -n "$v pe=$pe $mode: "
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --plant-every $pe --mode $mode 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['roofline']['kernel_ms_avg'], j['roofline']['kernel_ms_min'], j['config']['kernel'], j['summary']['records_hit'])" || exit 1
  done; done
done; done > gpurun_out/r02_flaglist_ab.txt 2>&1
cat gpurun_out/r02_flaglist_ab.txt

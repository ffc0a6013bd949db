#!/bin/bash
# LLVM AMDGPU scheduling strategies for the scan kernels, same box: tools/sched_ab.sh
# (builds: python -m merkurio_amd.build --tag smaxilp --flags "-mllvm -amdgpu-sched-strategy=max-ilp" etc.)
cd ${GRAFT_REPO_ROOT:-/root/repo}
L=$PWD/merkurio_amd/lib
for r in 1 2; do for v in base smaxilp smaxmemoryclause sgcniterativeilp; do
  if [ $v = base ]; then unset MERKURIO_LIB_PATH; else export MERKURIO_LIB_PATH=$L/libmerkurio_hip_$v.so; fi
  for pe in 0 100; do
  echo -n "$v pe=$pe: "
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --plant-every $pe 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['roofline']['kernel_ms_avg'], j['roofline']['kernel_ms_min'])" || exit 1
  done
done; done

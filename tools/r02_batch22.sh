#!/bin/bash
# which form of the 16-byte comparison costs the no-hit scan least: A/B/C/D on one box
set -o pipefail
mkdir -p gpurun_out
L=$PWD/merkurio_amd/lib
for pe in 0 10 1; do for mode in any hits; do for r in 1 2; do for v in cmp8 new xv1 xv2; do
  if [ $v = new ]; then unset MERKURIO_LIB_PATH; else export MERKURIO_LIB_PATH=$L/libmerkurio_hip_$v.so; fi
  echo -n "pe=$pe mode=$mode $v: "
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 8 --warmup 2 --plant-every $pe --mode $mode 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['roofline']['kernel_ms_avg'], j['summary']['hits'])" || exit 1
done; done; done; done 2>&1 | tee gpurun_out/b22_ab.txt

#!/bin/bash
# 16-byte loads in the level-3 comparison: parity first, then A/B against the 8-byte build
# (merkurio_amd/lib/libmerkurio_hip_cmp8.so) at several hit rates, both modes
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/b21_tests.log 2>&1 || { tail -30 gpurun_out/b21_tests.log; exit 1; }
tail -2 gpurun_out/b21_tests.log
for pe in 0 10 1; do for mode in any hits; do for r in 1 2; do for v in old new; do
  if [ $v = old ]; then export MERKURIO_LIB_PATH=$PWD/merkurio_amd/lib/libmerkurio_hip_cmp8.so; else unset MERKURIO_LIB_PATH; fi
  echo -n "pe=$pe mode=$mode $v: "
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 8 --warmup 2 --plant-every $pe --mode $mode 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['roofline']['kernel_ms_avg'], j['config']['kernel'])" || exit 1
done; done; done; done 2>&1 | tee gpurun_out/b21_ab.txt

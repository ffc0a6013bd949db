#!/bin/bash
# A/B of the smaller configurations (C2 shape, C4 shape, C5 shard) against the previous build
cd ${GRAFT_REPO_ROOT:-/root/repo}
L=$PWD/merkurio_amd/lib
run() { timeout -k 10 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['roofline']['kernel_ms_avg'], j['roofline']['kernel_ms_min'], j['config']['kernel'])"; }
for r in 1 2 3; do for v in prev new; do
  if [ $v = new ]; then unset MERKURIO_LIB_PATH; else export MERKURIO_LIB_PATH=$L/libmerkurio_hip_$v.so; fi
  echo -n "$v C2: "; run --records 10000000 --patterns 1024 --rc --steps 20
  echo -n "$v C4: "; run --records 20000000 --patterns 10000 --steps 20 --mode hits
  echo -n "$v C5: "; run --records 12500000 --read-len 250 --patterns 500000 --k 21 --steps 5
done; done

#!/bin/bash
# A/B of alternative library builds on the same box: tools/ab3.sh "<bench args>" <alt1.so> [<alt2.so> ...]
ARGS=$1; shift
for r in 1 2 3; do
  unset MERKURIO_LIB_PATH
  echo -n "default: "; timeout -k 10 300 python bench.py --no-cpu-baseline --steps 10 $ARGS 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms_avg'])"
  for alt in "$@"; do
    export MERKURIO_LIB_PATH=$alt
    echo -n "$(basename $alt): "; timeout -k 10 300 python bench.py --no-cpu-baseline --steps 10 $ARGS 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms_avg'])"
  done
done

#!/bin/bash
# sparse (non-temporal stream) vs hit-dense (cacheable stream, early probe / drain) kernel flavour by hit rate:
# where should the host switch?  (--density-hint pins the flavour)
run() { echo -n "$1: "; shift; timeout -k 10 300 python bench.py --no-cpu-baseline --no-other-configs --steps 5 --warmup 2 "$@" 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('kernel', j['roofline']['kernel_ms_avg'], 'ms', j['config']['kernel'])"; }
for pe in 50 20 12 10 8 6 5 4; do for h in 0 1000; do for mode in any hits; do
  run "1 in $pe reads hit, hint $h, $mode" --plant-every $pe --density-hint $h --mode $mode --no-order
done; done; done

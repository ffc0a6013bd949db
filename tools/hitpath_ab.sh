#!/bin/bash
# where does the time of a hit-heavy scan go?  every read (or 1 in 3) contains a pattern
for pe in 1 3; do for extra in "" "--no-counters"; do for mode in any hits; do
echo -n "plant_every=$pe mode=$mode $extra: "
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 5 --warmup 2 --plant-every $pe --mode $mode $extra 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms_avg'])"
done; done; done

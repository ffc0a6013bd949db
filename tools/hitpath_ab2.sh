#!/bin/bash
# hit-heavy scans: default build vs an alternative build ($1)
ALT=$1
for pe in 1 3 100; do for v in new alt; do
if [ $v = alt ]; then export MERKURIO_LIB_PATH=$ALT; else unset MERKURIO_LIB_PATH; fi
for mode in any hits; do
echo -n "$v plant_every=$pe mode=$mode: "
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 5 --warmup 2 --plant-every $pe --mode $mode 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms_avg'])"
done; done; done

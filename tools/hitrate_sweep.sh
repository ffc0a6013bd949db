#!/bin/bash
# headline workload at different fractions of reads that really contain a pattern (1 in N), flags and hits mode
for pe in 0 1000 100 10 3 1; do for mode in any hits; do
echo -n "plant_every=$pe mode=$mode: "
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 5 --warmup 2 --plant-every $pe --mode $mode 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms_avg'], j['config']['kernel'], j['summary']['hits'], j['summary']['filter_candidates'])"
done; done

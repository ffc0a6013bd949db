#!/bin/bash
# headline workload at different fractions of reads that really contain a pattern (1 in N), flags and hits mode.
# usage: tools/hitrate_sweep.sh ["extra bench.py args"]   e.g. "--with-offsets" for the record lookup through the offsets array
EXTRA="$1"
for pe in 0 1000 100 10 3 1; do for mode in any hits; do
echo -n "plant_every=$pe mode=$mode $EXTRA: "
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 5 --warmup 2 --plant-every $pe --mode $mode $EXTRA 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('step', j['ms_per_step'], 'ms; scan kernel', r['kernel_ms_avg'], 'ms; order', r.get('order_ms_avg','-'), 'ms;', j['config']['kernel'], 'hits', j['summary']['hits']//5, 'candidates', j['summary']['filter_candidates']//5)"
done; done

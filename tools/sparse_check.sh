#!/bin/bash
# sparse-candidate sanity on the 15 GB batch: with 1 read in N planted, about 100 M / N reads must hit
for cfg in "1 1000" "1 100000" "13 1000" "10000 100000"; do set -- $cfg
echo -n "patterns=$1 plant_every=$2 (expect ~$((100000000 / $2))): "
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --patterns $1 --plant-every $2 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['summary']['hits']//2, j['summary']['records_hit']//2, j['summary']['filter_candidates']//2, j['config']['kernel'])"
done

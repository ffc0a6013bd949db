#!/bin/bash
# config-5 shard with stride 8 (q = 14) against the default stride 4 (q = 18), several filter sizes
for cfg in "4 18" "8 18" "8 19" "8 20" "8 21"; do set -- $cfg
echo -n "stride=$1 log2_blocks=$2: "
timeout -k 10 300 python bench.py --no-cpu-baseline --records 12500000 --read-len 250 --patterns 500000 --k 21 --steps 5 --gbloom-log2-blocks $2 --force-stride $1 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['roofline']['kernel_ms_avg'], j['config']['kernel'], j['config']['filter']['filter_bytes'], j['summary']['filter_candidates'], j['summary']['hits'])"
done

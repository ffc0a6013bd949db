#!/bin/bash
# config-5 shard: A/B of an alternative library build ($1) against the default one, strides 4 and 8
ALT=$1
for r in 1 2; do for cfg in "4 18" "8 19"; do set -- $cfg; for v in new alt; do
if [ $v = alt ]; then export MERKURIO_LIB_PATH=$ALT; else unset MERKURIO_LIB_PATH; fi
echo -n "$v stride=$1 log2_blocks=$2: "
timeout -k 10 300 python bench.py --no-cpu-baseline --records 12500000 --read-len 250 --patterns 500000 --k 21 --steps 5 --gbloom-log2-blocks $2 --force-stride $1 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['roofline']['kernel_ms_avg'], j['config']['kernel'], j['summary']['filter_candidates'], j['summary']['hits'])"
done; done; done

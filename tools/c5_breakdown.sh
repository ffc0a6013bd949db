#!/bin/bash
# config-5 shard: full vs -DMK_ABLATE=1 (filter probes made, candidates dropped) vs -DMK_ABLATE=7 (loads + pack)
L=$PWD/merkurio_amd/lib
for cfg in "4 18" "8 19"; do set -- $cfg; for v in full abl1 abl7; do
if [ $v = full ]; then unset MERKURIO_LIB_PATH; else export MERKURIO_LIB_PATH=$L/libmerkurio_hip_$v.so; fi
echo -n "$v stride=$1 log2_blocks=$2: "
timeout -k 10 300 python bench.py --no-cpu-baseline --records 12500000 --read-len 250 --patterns 500000 --k 21 --steps 5 --gbloom-log2-blocks $2 --force-stride $1 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['roofline']['kernel_ms_avg'], j['config']['kernel'], j['summary']['filter_candidates'], j['summary']['hits'])"
done; done

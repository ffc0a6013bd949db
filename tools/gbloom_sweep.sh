for lb in 17 18 19 20 21 22; do for s in 4 2; do
echo -n "log2_blocks=$lb stride=$s: "
timeout -k 10 300 python bench.py --no-cpu-baseline --records 12500000 --read-len 250 --patterns 500000 --k 21 --steps 5 --gbloom-log2-blocks $lb --force-stride $s 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['roofline']['kernel_ms_avg'], j['config']['kernel'], j['config']['filter']['filter_bytes'], j['summary']['filter_candidates'])"
done; done

run() { echo -n "$*: "; timeout -k 10 300 python bench.py --no-cpu-baseline --steps 10 "$@" 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms_avg'], j['config']['kernel'], j['summary']['filter_candidates'])"; }
run --patterns 13
run --patterns 13 --plant-every 0
run --patterns 13 --no-counters
run --patterns 2048
run --patterns 2048 --plant-every 0
run --patterns 10000

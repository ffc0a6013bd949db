run() { echo -n "$1: "; shift; timeout -k 10 300 python bench.py --no-cpu-baseline --steps 10 "$@" 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms_avg'])"; }
for r in 1 2; do
run "counters   "
run "no counters" --no-counters
done

for s in 2 4 8 16; do
  echo "== --force-stride $s"
  timeout -k 10 200 python bench.py --records 40000000 --steps 5 --warmup 1 --no-cpu-baseline --force-stride $s 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms_avg'], j['summary']['filter_candidates'], j['config']['filter'], j['summary']['hits'])"
done

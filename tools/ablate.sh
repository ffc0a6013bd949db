for d in 0 1 3 7; do
  echo "== MERKURIO_DEBUG=$d"
  MERKURIO_DEBUG=$d timeout -k 10 200 python bench.py --records 40000000 --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['roofline']['kernel_ms_avg'], j['summary']['filter_candidates'])"
done

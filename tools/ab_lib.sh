# A/B two builds of the library on the same box: tools/ab_lib.sh <old.so> [bench args]
OLD=$1; shift
for r in 1 2 3; do
for v in old new; do
  if [ $v = old ]; then export MERKURIO_LIB_PATH=$OLD; else unset MERKURIO_LIB_PATH; fi
  echo -n "$v: "
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 10 "$@" 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms_avg'])"
done; done

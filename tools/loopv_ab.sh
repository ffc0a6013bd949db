for v in 0 1 0 1; do
  MERKURIO_HIPCC_FLAGS="-DMK_LOOPV=$v" python -m merkurio_amd.build --force > /dev/null 2>&1
  for s in 8 16; do
  echo -n "MK_LOOPV=$v stride=$s: "
  MERKURIO_FORCE_STRIDE=$s timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms_avg'])"
  done
done
python -m merkurio_amd.build --force > /dev/null 2>&1

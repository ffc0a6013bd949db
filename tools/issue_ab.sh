for v in 16 32 48 56 32; do
  MERKURIO_HIPCC_FLAGS="-DMK_ISSUE_AT=$v" python -m merkurio_amd.build --force > /dev/null 2>&1
  echo -n "MK_ISSUE_AT=$v: "
  timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms_avg'])"
done
python -m merkurio_amd.build --force > /dev/null 2>&1

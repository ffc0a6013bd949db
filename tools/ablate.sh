# compile-time ablation of the scan kernel (runs on the GPU box; rebuilds the library in place)
# usage: tools/ablate.sh [stride] [records]
S=${1:-0}; R=${2:-100000000}
for a in 0 1 3 7 0; do
  MERKURIO_HIPCC_FLAGS="-DMK_ABLATE=$a" python -m merkurio_amd.build --force > /dev/null 2>&1
  echo -n "MK_ABLATE=$a stride=$S: "
  timeout -k 10 200 python bench.py --records $R --steps 10 --warmup 2 --no-cpu-baseline --force-stride $S 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms_avg'], j['summary']['filter_candidates'], j['config']['kernel'])"
done
python -m merkurio_amd.build --force > /dev/null 2>&1

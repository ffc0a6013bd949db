# two-class kernel by ablation on one box (after: python -m merkurio_amd.build --tag mcabl1 --flags "-DMK_MC_ABL=1"; --tag mcabl2
# --flags "-DMK_MC_ABL=2"): the default build, the kernel without its short-class samples, with one compile-time geometry
for r in 1 2; do
for v in default mcabl1 mcabl2; do
  if [ $v = default ]; then unset MERKURIO_LIB_PATH; else export MERKURIO_LIB_PATH=merkurio_amd/lib/libmerkurio_hip_$v.so; fi
  echo "== $v"
  timeout -k 10 300 python tools/mixed_sets.py --only headline,plus8 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    n,j=l.split(' ',1); j=json.loads(j); print(' ',n,j['kernel_ms'],j['frac'],j['candidates'],j['hits'])"
done; done

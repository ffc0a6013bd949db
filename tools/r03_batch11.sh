#!/bin/bash
# round 3, batch 11: leaf sort launched per size class: ordering tests, ordering timings, hits-mode lines
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_order.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py -x -q -m gpu -k "order or fuzz or every_read or 2_pow_32" 2>&1 | tail -3 | tee gpurun_out/r03_pytest_b11.txt
rm -f gpurun_out/r03_order_hits.txt
for n in 1000000 4000000 100000000; do
  timeout -k 10 300 python tools/order_hits_bench.py $n 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03_order_hits.txt
done
{ echo "# C4 shape (20 M x 150 bp, 10 k 31-mers, hit tuples, ordered)"; timeout -k 10 300 python bench.py --records 20000000 --mode hits --no-cpu-baseline --steps 10 2>&1 | tail -1
  echo "# every read hits (100 M x 150 bp), tuples ordered"; timeout -k 10 300 python bench.py --mode hits --plant-every 1 --no-cpu-baseline --steps 5 2>&1 | tail -1
  echo "# 10 % of the reads hit"; timeout -k 10 300 python bench.py --mode hits --plant-every 10 --no-cpu-baseline --steps 5 2>&1 | tail -1; } > gpurun_out/r03_hits_mode.txt
python - <<'PY'
import json
for l in open('gpurun_out/r03_hits_mode.txt'):
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; print(j['ms_per_step'], r['kernel_ms_avg'], r['frac'], r['order_ms_avg'], r['frac_scan_plus_order'], r['order'])
PY

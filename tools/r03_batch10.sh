#!/bin/bash
# round 3, batch 10: leaf sort with wave-local rounds (no workgroup barrier between them): ordering tests + timings
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_order.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/r03_pytest_b10.txt
rm -f gpurun_out/r03_order_hits_b10.txt
for n in 1000000 4000000 100000000; do
  timeout -k 10 300 python tools/order_hits_bench.py $n 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03_order_hits_b10.txt
done

#!/bin/bash
# round 3, batch 1: the hand-written emission-order kernels -- parity first, then timings
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_order.py tests/test_gpu_parity.py -x -q -m gpu -k "order" 2>&1 | tail -15
for n in 1000000 4000000 100000000; do
  timeout -k 10 300 python tools/order_hits_bench.py $n 2>&1 | tee -a gpurun_out/r03_order_hits.txt
done

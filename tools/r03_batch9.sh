#!/bin/bash
# round 3, batch 9: paired extract on the device: full GPU suite + a fuzz campaign over the new driver loops
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/r03_pytest_b9.txt
timeout -k 10 600 bash tools/fuzz_campaign.sh 200 7000 2>&1 | tail -5 | tee gpurun_out/r03_fuzz_campaign.txt
timeout -k 10 300 python bench.py 2>&1 | tail -1 > gpurun_out/r03_bench_default_b9.json

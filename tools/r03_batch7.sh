#!/bin/bash
# round 3, batch 7: the driver loops on the device (extract single, tag): full GPU suite, then where a dense call
# spends its time
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 | tee gpurun_out/r03_pytest_b7.txt
timeout -k 10 300 python tools/tag_records_bench.py 4000000 2>&1 | tee gpurun_out/r03_tag_records.txt

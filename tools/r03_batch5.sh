#!/bin/bash
# round 3, batch 5: full GPU suite (new: > 2^32 records, > 4 GiB record, fixed-length batches), the default bench
# line with other_configs, the reference's own benchmark shape end to end
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 | tee gpurun_out/r03_pytest_b5.txt
(time timeout -k 10 500 python bench.py) 2>&1 | tail -5 | tee gpurun_out/r03_bench_default_b5.json
export TMPDIR=/tmp
timeout -k 10 500 python tools/refbench.py 10 2>&1 | tee gpurun_out/r03_refbench.txt

#!/bin/bash
# round 3, batch 6: the two tests that failed in batch 5 (test-side fixes), the reference's benchmark shape,
# first judged profiles of the round (headline + config-5 shard)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "2_pow_32 or beyond_4_gib or config4_full" 2>&1 | tail -12 | tee gpurun_out/r03_pytest_b6.txt
export TMPDIR=/tmp
timeout -k 10 500 python tools/refbench.py 10 2>&1 | tee gpurun_out/r03_refbench.txt
bash tools/profile_gpu.sh r03 100000000 5 > gpurun_out/r03_profile.log 2>&1
bash tools/profile_gpu.sh r03_c5 12500000 5 --read-len 250 --patterns 500000 --k 21 > gpurun_out/r03_profile_c5.log 2>&1
tail -30 gpurun_out/prof_r03/summary.txt

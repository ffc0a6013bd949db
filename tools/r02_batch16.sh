#!/bin/bash
# r02 GPU batch 16: judged profile + default bench line on the final kernel source; pattern-set compilation times
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -2
bash tools/profile_gpu.sh r02_headline 100000000 5 > $O/r02_profile_headline.log 2>&1; grep -E "mk_scan_kernel|FETCH|WRITE" $O/prof_r02_headline/summary.txt | cut -c1-220
cd ${GRAFT_REPO_ROOT:-/root/repo}
python bench.py > $O/r02_bench_default.json 2> $O/r02_bench_default.err; python -c "
import json; j=json.load(open('$O/r02_bench_default.json')); print(j['value'], j['roofline'])"
python tools/compile_time.py > $O/r02_compile_time.txt 2>&1; cat $O/r02_compile_time.txt

#!/bin/bash
# r02 GPU batch 15: judged profile of the headline (kernel trace at the bench's own step counts + PMC passes) and the
# default bench line, with the traffic figure of the committed profile now matching the kernel source
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
bash tools/profile_gpu.sh r02_headline 100000000 5 > $O/r02_profile_headline.log 2>&1; grep -E "mk_scan_kernel|FETCH|WRITE" $O/prof_r02_headline/summary.txt | cut -c1-220
cd ${GRAFT_REPO_ROOT:-/root/repo}
python bench.py > $O/r02_bench_default.json 2> $O/r02_bench_default.err; python -c "
import json; j=json.load(open('$O/r02_bench_default.json')); print(j['value'], j['roofline'])"

#!/bin/bash
# r02 GPU batch 17: the default bench line next to its rocprofv3 kernel trace, same box, final kernel source
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_r02_final/kt -o kt -- python3 $R/bench.py --no-cpu-baseline > $R/$O/prof_r02_final.log 2>&1
grep mk_scan_kernel $R/$O/prof_r02_final/kt/kt_kernel_stats.csv | cut -c1-200
cd $R
python bench.py > $O/r02_bench_default.json 2> $O/r02_bench_default.err; python -c "
import json; j=json.load(open('$O/r02_bench_default.json')); print(j['value'], j['roofline'])"

#!/bin/bash
# r02 GPU batch 25: judged profile + default bench line, same box (batch 24 without the sweep)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out
cd $R
bash tools/profile_gpu.sh r02_headline 100000000 5 > $O/r02_profile_headline.log 2>&1 || exit 1
grep -E "mk_scan_kernel|FETCH|WRITE" $O/prof_r02_headline/summary.txt | cut -c1-200
cp $O/prof_r02_headline/traffic.json profiles/traffic_r02.json   # box-local: the bench below may then quote it
bash tools/r02_batch17.sh

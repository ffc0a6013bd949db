#!/bin/bash
# r02 GPU batch 35: full suite, judged profile + default bench line on one box, hit-rate sweep, config sweep
R=${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out
cd $R
( time timeout -k 10 900 python -m pytest tests -m gpu -x -q ) > $O/r02_gputest35.log 2>&1 || { tail -30 $O/r02_gputest35.log; exit 1; }
tail -4 $O/r02_gputest35.log
bash tools/profile_gpu.sh r02_headline 100000000 5 > $O/r02_profile_headline.log 2>&1 || exit 1
grep -E "mk_scan_kernel|FETCH|WRITE" $O/prof_r02_headline/summary.txt | cut -c1-200
cp $O/prof_r02_headline/traffic.json profiles/traffic_r02.json   # box-local: the bench below may then quote it
bash tools/r02_batch17.sh || exit 1
cd $R
bash tools/hitrate_sweep.sh > $O/r02_hitrate_sweep.txt 2>&1; cat $O/r02_hitrate_sweep.txt

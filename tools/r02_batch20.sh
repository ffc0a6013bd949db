#!/bin/bash
# r02 GPU batch 20: coarse record index for ragged batches: tests, ragged sweep, judged profile on the new kernel source
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
( time python -m pytest tests -m gpu -x -q --durations=3 ) > $O/r02_gputest20.log 2>&1; echo "pytest rc=$?" >> $O/r02_gputest20.log; tail -8 $O/r02_gputest20.log
bash tools/ragged_sweep.sh > $O/r02_ragged_sweep.txt 2>&1; cat $O/r02_ragged_sweep.txt
bash tools/profile_gpu.sh r02_headline 100000000 5 > $O/r02_profile_headline.log 2>&1; grep -E "mk_scan_kernel|FETCH|WRITE" $O/prof_r02_headline/summary.txt | cut -c1-220

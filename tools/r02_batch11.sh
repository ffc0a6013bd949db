#!/bin/bash
# r02 GPU batch 11: tests + end-to-end CLI timings after dropping MAP_POPULATE (windows are requested with madvise)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
( time python -m pytest tests -m gpu -x -q --durations=4 ) > $O/r02_gputest11.log 2>&1; echo "pytest rc=$?" >> $O/r02_gputest11.log; tail -10 $O/r02_gputest11.log
export TMPDIR=/tmp
python tools/e2e_cli.py 20000000 > $O/r02_e2e_extract2.txt 2>&1; grep -v "batch:\|next window" $O/r02_e2e_extract2.txt | tail -12
python tools/e2e_tag.py 2000000 > $O/r02_e2e_tag2.txt 2>&1; grep -v "batch:\|timing" $O/r02_e2e_tag2.txt | tail -8

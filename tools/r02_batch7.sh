#!/bin/bash
# r02 GPU batch 7: tests with the windowed host program, clean size / config sweeps, end-to-end CLI timings
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
( time python -m pytest tests -m gpu -x -q --durations=6 ) > $O/r02_gputest7.log 2>&1; echo "pytest rc=$?" >> $O/r02_gputest7.log; tail -12 $O/r02_gputest7.log
bash tools/size_sweep.sh > $O/r02_size_sweep3.txt 2>&1; cat $O/r02_size_sweep3.txt
bash tools/config_sweep.sh > $O/r02_config_sweep3.txt 2>&1; cat $O/r02_config_sweep3.txt
export TMPDIR=/tmp
python tools/e2e_cli.py 20000000 > $O/r02_e2e_extract.txt 2>&1; grep -v "batch:" $O/r02_e2e_extract.txt | tail -12
python tools/e2e_tag.py 2000000 > $O/r02_e2e_tag.txt 2>&1; grep -v "batch:" $O/r02_e2e_tag.txt | tail -14

#!/bin/bash
# r02 GPU batch 14: final kernel source (record-index guess nudge; aging back at 1 GiB): tests, judged profiles, sweeps
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
( time python -m pytest tests -m gpu -x -q --durations=4 ) > $O/r02_gputest14.log 2>&1; echo "pytest rc=$?" >> $O/r02_gputest14.log; tail -9 $O/r02_gputest14.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/r02_smoke.log 2>&1; tail -1 $O/r02_smoke.log
bash tools/profile_gpu.sh r02_headline 100000000 5 > $O/r02_profile_headline.log 2>&1; tail -4 $O/r02_profile_headline.log | cut -c1-300
bash tools/profile_gpu.sh r02_c5 12500000 5 --read-len 250 --patterns 500000 --k 21 > $O/r02_profile_c5.log 2>&1; tail -2 $O/r02_profile_c5.log | cut -c1-300
cd ${GRAFT_REPO_ROOT:-/root/repo}
python bench.py > $O/r02_bench_default.json 2> $O/r02_bench_default.err; cat $O/r02_bench_default.json | cut -c1-400
bash tools/hitrate_sweep.sh > $O/r02_hitrate_sweep6.txt 2>&1; cat $O/r02_hitrate_sweep6.txt
bash tools/config_sweep.sh > $O/r02_config_sweep4.txt 2>&1; cut -c1-200 $O/r02_config_sweep4.txt
bash tools/size_sweep.sh > $O/r02_size_sweep4.txt 2>&1; cat $O/r02_size_sweep4.txt

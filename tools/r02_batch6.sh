#!/bin/bash
# r02 GPU batch 6: tests on the final kernel source, then the judged profiles: rocprofv3 kernel trace +
# PMC passes of the default bench command (headline) and of the config-5 shard, and the default bench line
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
( time python -m pytest tests -m gpu -x -q --durations=5 ) > $O/r02_gputest6.log 2>&1; echo "pytest rc=$?" >> $O/r02_gputest6.log; tail -8 $O/r02_gputest6.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/r02_smoke.log 2>&1; tail -2 $O/r02_smoke.log
bash tools/profile_gpu.sh r02_headline 100000000 5 > $O/r02_profile_headline.log 2>&1; tail -45 $O/r02_profile_headline.log
bash tools/profile_gpu.sh r02_c5 12500000 5 --read-len 250 --patterns 500000 --k 21 > $O/r02_profile_c5.log 2>&1; tail -30 $O/r02_profile_c5.log
cd ${GRAFT_REPO_ROOT:-/root/repo}
python bench.py > $O/r02_bench_default.json 2> $O/r02_bench_default.err; cat $O/r02_bench_default.json

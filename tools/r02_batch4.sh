#!/bin/bash
# r02 GPU batch 4: tests; config + size sweeps after the per-workgroup counter reduction, run-length tile
# dealing and the 3 MiB four-bit global filter; tile-run and hit-rate checks
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
L=merkurio_amd/lib
one() { timeout -k 10 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; s=j['summary']; print(j['value'], 'Gbases/s ms/step', j['ms_per_step'], 'kernel avg/min', r['kernel_ms_avg'], r['kernel_ms_min'], 'frac', r['frac'], j['config']['kernel'], 'hits', s['hits'], 'cand', s['filter_candidates'])"; }
( time python -m pytest tests -m gpu -x -q --durations=8 ) > $O/r02_gputest4.log 2>&1; echo "pytest rc=$?" >> $O/r02_gputest4.log; tail -16 $O/r02_gputest4.log
bash tools/size_sweep.sh > $O/r02_size_sweep2.txt 2>&1; cat $O/r02_size_sweep2.txt
bash tools/config_sweep.sh > $O/r02_config_sweep2.txt 2>&1; cat $O/r02_config_sweep2.txt
{
echo "# tile_run at the headline size, two rounds"
for r in 1 2; do for tr in 1 2 4 8; do echo -n "headline tile_run=$tr: "; one --steps 10 --tile-run $tr; done; done
echo "# config-5 shard: filter size with the four-bit filter"
C5="--records 12500000 --read-len 250 --patterns 500000 --k 21 --steps 5"
for kib in 2560 3072 3584 4096; do echo -n "kib=$kib: "; one $C5 --gbloom-kib $kib; done
echo -n "default: "; one $C5
echo -n "default, hits mode: "; one $C5 --mode hits
} > $O/r02_tile_c5.txt 2>&1; cat $O/r02_tile_c5.txt
bash tools/hitrate_sweep.sh > $O/r02_hitrate_sweep3.txt 2>&1; cat $O/r02_hitrate_sweep3.txt

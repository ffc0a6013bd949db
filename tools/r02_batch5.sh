#!/bin/bash
# r02 GPU batch 5: box clocks, tests, headline A/B (stride 8 vs 16, tile runs), hit-rate sweep with the
# density-steered load flavour, C2 / C5
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
one() { timeout -k 10 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; s=j['summary']; print(j['value'], 'Gbases/s ms/step', j['ms_per_step'], 'kernel avg/min', r['kernel_ms_avg'], r['kernel_ms_min'], 'frac', r['frac'], j['config']['kernel'], 'hits', s['hits'], 'cand', s['filter_candidates'])"; }
{ rocm-smi --showclocks --showpower --showtemp 2>&1 | head -40; } > $O/r02_box5.txt 2>&1
( time python -m pytest tests -m gpu -x -q --durations=5 ) > $O/r02_gputest5.log 2>&1; echo "pytest rc=$?" >> $O/r02_gputest5.log; tail -12 $O/r02_gputest5.log
{
echo "# headline: stride 8 (q 24) vs stride 16 (q 16); tile runs; two rounds"
for r in 1 2; do
  echo -n "default: "; one --steps 10
  echo -n "stride 16: "; one --steps 10 --force-stride 16
  echo -n "tile_run 1: "; one --steps 10 --tile-run 1
  echo -n "tile_run 4: "; one --steps 10 --tile-run 4
  echo -n "13 patterns: "; one --steps 10 --patterns 13
done
( rocm-smi --showclocks 2>&1 | grep -i -E "sclk|mclk|fclk" | head -8 )
echo -n "C2: "; one --records 10000000 --patterns 1024 --rc --steps 20
echo -n "C2 stride 8: "; one --records 10000000 --patterns 1024 --rc --steps 20 --force-stride 8
echo -n "C4 shape hits: "; one --records 20000000 --steps 20 --mode hits
echo -n "C5: "; one --records 12500000 --read-len 250 --patterns 500000 --k 21 --steps 5
} > $O/r02_headline_ab.txt 2>&1; cat $O/r02_headline_ab.txt
bash tools/hitrate_sweep.sh > $O/r02_hitrate_sweep4.txt 2>&1; cat $O/r02_hitrate_sweep4.txt

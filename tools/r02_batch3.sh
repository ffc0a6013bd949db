#!/bin/bash
# r02 GPU batch 3: tests (incl. the 20 M-record BAM case), stream cache-policy probe, tile-run A/B,
# small-batch ablations, config-5 filter sizes, hit-rate sweep after moving the per-pattern counts out
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
L=merkurio_amd/lib
one() { timeout -k 10 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; s=j['summary']; print(j['value'], 'Gbases/s ms/step', j['ms_per_step'], 'kernel avg/min', r['kernel_ms_avg'], r['kernel_ms_min'], 'frac', r['frac'], j['config']['kernel'], 'hits', s['hits'], 'cand', s['filter_candidates'])"; }
( time python -m pytest tests -m gpu -x -q --durations=8 ) > $O/r02_gputest3.log 2>&1; echo "pytest rc=$?" >> $O/r02_gputest3.log; tail -16 $O/r02_gputest3.log
./tools/probes/stream_policy > $O/r02_stream_policy.txt 2>&1; cat $O/r02_stream_policy.txt
{
echo "# tile_run A/B (consecutive tiles per wave before the jump): headline, C2 shape, 1 M reads"
for tr in 1 2 4 8; do
  echo -n "headline tile_run=$tr: "; one --steps 10 --tile-run $tr
  echo -n "C2 tile_run=$tr: "; one --records 10000000 --patterns 1024 --rc --steps 20 --tile-run $tr
  echo -n "1M reads tile_run=$tr: "; one --records 1000000 --steps 50 --warmup 5 --tile-run $tr
done
} > $O/r02_tile_run.txt 2>&1; cat $O/r02_tile_run.txt
{
echo "# small batches, ablation builds (abl7 = loads + pack only, abl1 = + filter, positives dropped)"
for n in 1000000 2500000 10000000; do for lib in main abl1 abl7; do
  echo -n "records=$n lib=$lib: "
  if [ $lib = main ]; then one --records $n --steps 30 --warmup 5; else MERKURIO_LIB_PATH=$L/libmerkurio_hip_$lib.so one --records $n --steps 30 --warmup 5; fi
done; done
echo -n "records=1000000 main plant-every 0: "; one --records 1000000 --steps 30 --warmup 5 --plant-every 0
echo -n "records=1000000 main no-counters: "; one --records 1000000 --steps 30 --warmup 5 --no-counters
} > $O/r02_small_ablate.txt 2>&1; cat $O/r02_small_ablate.txt
{
echo "# config-5 shard, global filter size (KiB); full build and abl1 (probes made, candidates dropped)"
C5="--records 12500000 --read-len 250 --patterns 500000 --k 21 --steps 5"
for kib in 2048 2560 3072 3584 4096 5120; do
  echo -n "kib=$kib full: "; one $C5 --gbloom-kib $kib
  echo -n "kib=$kib abl1: "; MERKURIO_LIB_PATH=$L/libmerkurio_hip_abl1.so one $C5 --gbloom-kib $kib
done
} > $O/r02_c5_filter_size.txt 2>&1; cat $O/r02_c5_filter_size.txt
{
echo "# headline workload, 1 read in N planted; per-pattern counts now come from a histogram of the tuples (hits mode)"
for pe in 0 100 10 3 1; do for lib in main adnt; do for mode in any hits; do
  echo -n "plant_every=$pe lib=$lib mode=$mode: "
  if [ $lib = main ]; then one --steps 5 --warmup 2 --plant-every $pe --mode $mode; else MERKURIO_LIB_PATH=$L/libmerkurio_hip_adnt.so one --steps 5 --warmup 2 --plant-every $pe --mode $mode; fi
done; done; done
} > $O/r02_hitrate_sweep2.txt 2>&1; cat $O/r02_hitrate_sweep2.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 -L > $R/$O/r02_counters_avail.txt 2>&1
grep -i -E "utcl|tlb|TCC_HIT|TCC_MISS|TCC_REQ|TCP_TCC" $R/$O/r02_counters_avail.txt | head -40
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $R/$O/prof_c5_tcc -o pmc -- python3 $R/bench.py --records 12500000 --read-len 250 --patterns 500000 --k 21 --steps 3 --warmup 1 --no-cpu-baseline > $R/$O/prof_c5_tcc.log 2>&1
python3 - <<PY
import csv,glob
from collections import defaultdict
for f in glob.glob("$R/$O/prof_c5_tcc/**/*counter_collection.csv", recursive=True):
    acc=defaultdict(list)
    for row in csv.DictReader(open(f)):
        if "mk_scan" in row["Kernel_Name"]: acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k,v in sorted(acc.items()): print("C5", k, len(v), sum(v)/len(v))
PY

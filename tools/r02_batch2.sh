#!/bin/bash
# r02 GPU batch 2: tests, config-5 breakdown of the reworked global-filter kernel, hit-rate sweep with
# plain vs adaptive stream loads, fixed per-launch overhead probe (events vs rocprofv3 kernel trace)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
L=merkurio_amd/lib
one() { timeout -k 10 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; s=j['summary']; print(j['value'], 'Gbases/s kernel avg/min', r['kernel_ms_avg'], r['kernel_ms_min'], 'frac', r['frac'], j['config']['kernel'], 'hits', s['hits'], 'cand', s['filter_candidates'])"; }
python -m pytest tests -m gpu -x -q > $O/r02_gputest2.log 2>&1; echo "pytest rc=$?" >> $O/r02_gputest2.log; tail -4 $O/r02_gputest2.log
{
echo "# tools/r02_batch2.sh: config-5 shard (12.5 M x 250 bp, 500 k 21-mers), reworked global-filter kernel (context fingerprints, probes one chunk ahead, table load 0.25)"
C5="--records 12500000 --read-len 250 --patterns 500000 --k 21 --steps 5"
echo -n "full: "; one $C5
echo -n "abl1 (probes made, candidates dropped): "; MERKURIO_LIB_PATH=$L/libmerkurio_hip_abl1.so one $C5
echo -n "abl7 (loads + pack only): "; MERKURIO_LIB_PATH=$L/libmerkurio_hip_abl7.so one $C5
echo -n "full, no planted hits: "; one $C5 --plant-every 0
echo -n "full, hits mode: "; one $C5 --mode hits
echo -n "stride 4 (q 18, 2 MiB filter): "; one $C5 --force-stride 4
echo -n "log2_blocks 18 (2 MiB filter): "; one $C5 --gbloom-log2-blocks 18
echo -n "log2_blocks 20 (8 MiB filter): "; one $C5 --gbloom-log2-blocks 20
} > $O/r02_c5_breakdown.txt 2>&1
cat $O/r02_c5_breakdown.txt
{
echo "# tools/r02_batch2.sh: headline workload, 1 read in N planted; main build vs -DMK_ADAPTIVE_NT=1; columns as bench"
for pe in 0 100 10 3 1; do for lib in main adnt; do for cnt in "" "--no-counters"; do
  echo -n "plant_every=$pe lib=$lib $cnt: "
  if [ $lib = main ]; then one --steps 5 --warmup 2 --plant-every $pe $cnt; else MERKURIO_LIB_PATH=$L/libmerkurio_hip_adnt.so one --steps 5 --warmup 2 --plant-every $pe $cnt; fi
done; done; done
} > $O/r02_hitrate_sweep.txt 2>&1
cat $O/r02_hitrate_sweep.txt
{
echo "# tools/r02_batch2.sh: per-launch fixed cost: kernel time (hipEvents) for tiny batches; LDS filter vs global filter (no staging)"
for n in 1000 100000 1000000; do
  echo -n "records=$n lds: "; one --records $n --patterns 10000 --steps 50 --warmup 5
  echo -n "records=$n global-filter: "; one --records $n --patterns 10000 --steps 50 --warmup 5 --force-global-filter
done
} > $O/r02_overhead.txt 2>&1
cat $O/r02_overhead.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_small -o kt -- python3 $R/bench.py --records 1000000 --steps 50 --warmup 5 --no-cpu-baseline > $R/$O/prof_small.log 2>&1
python3 $R/tools/summarize_prof.py $R/$O/prof_small > $R/$O/r02_small_kernel_trace.txt 2>&1
cat $R/$O/r02_small_kernel_trace.txt | head -30

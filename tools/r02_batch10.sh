#!/bin/bash
# r02 GPU batch 10: instruction-cache behaviour of the scan kernel (its code is ~100 KB, the I-cache 64 KB per CU pair)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out
for pe in 0 100 10; do
  rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/$O/prof_ic_$pe -o pmc -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --plant-every $pe > $R/$O/prof_ic_$pe.log 2>&1
done
python3 - <<PY
import csv,glob
from collections import defaultdict
for pe in (0,100,10):
    for f in glob.glob("$R/$O/prof_ic_%d/**/*counter_collection.csv" % pe, recursive=True):
        acc=defaultdict(list)
        for row in csv.DictReader(open(f)):
            if "mk_scan" in row["Kernel_Name"]: acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        print("plant_every=%d"%pe, {k: "%.4g"%(sum(v)/len(v)) for k,v in sorted(acc.items())})
PY

#!/bin/bash
# r02 GPU batch 8: tests, end-to-end CLI timings with prefetched windows, HBM traffic of hit-heavy inputs
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
( time python -m pytest tests -m gpu -x -q --durations=4 ) > $O/r02_gputest8.log 2>&1; echo "pytest rc=$?" >> $O/r02_gputest8.log; tail -10 $O/r02_gputest8.log
export TMPDIR=/tmp
python tools/e2e_cli.py 20000000 > $O/r02_e2e_extract.txt 2>&1; grep -v "batch:" $O/r02_e2e_extract.txt | tail -24
python tools/e2e_tag.py 2000000 > $O/r02_e2e_tag.txt 2>&1; grep -v "batch:" $O/r02_e2e_tag.txt | tail -24
cd /tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for pe in 10 1; do for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$O/prof_hit_${pe}_$c -o pmc -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --plant-every $pe > $R/$O/prof_hit_${pe}_$c.log 2>&1
done; done
python3 - <<PY
import csv,glob
from collections import defaultdict
for pe in (10,1):
    for c in ("FETCH_SIZE","WRITE_SIZE"):
        for f in glob.glob("$R/$O/prof_hit_%d_%s/**/*counter_collection.csv" % (pe,c), recursive=True):
            acc=defaultdict(list)
            for row in csv.DictReader(open(f)):
                if "mk_scan" in row["Kernel_Name"]: acc[(row["Kernel_Name"][:60],row["Counter_Name"])].append(float(row["Counter_Value"]))
            for k,v in sorted(acc.items()): print("plant_every=%d"%pe, k, "n=%d"%len(v), "avg KB per launch = %.0f"%(sum(v)/len(v)), "last = %.0f"%v[-1])
PY

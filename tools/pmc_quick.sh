#!/bin/bash
# instruction / wait counters of the scan kernel for the library in $MERKURIO_LIB_PATH (or the default build)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pq
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d /tmp/pq -o pmc -- python3 $ROOT/bench.py --records 100000000 --steps 3 --warmup 1 --no-cpu-baseline "$@" > /tmp/pq.log 2>&1
python3 - <<PY
import csv, glob
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob("/tmp/pq/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "mk_scan" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("  {:20s} n={:3d} avg={:.6g}".format(k, len(v), sum(v) / len(v)))
PY

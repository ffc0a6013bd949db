#!/bin/bash
# HBM traffic and wave-cycle counters of a hit-heavy scan (every read carries a pattern) next to the default workload
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_hitpath
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for pe in ${PES:-100 1}; do
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/pe${pe}_pmc$i -o pmc -- python3 $ROOT/bench.py --records 100000000 --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --plant-every $pe > $OUT/pe${pe}_pmc$i.log 2>&1
  echo "plant_every=$pe pmc$i [$SET] rc=$?"
done
done
python3 - <<PY
import csv, glob, os
from collections import defaultdict
out = "$OUT"
for pe in [int(x) for x in "${PES:-100 1}".split()]:
    print("== plant_every=%d (per-launch averages, mk_scan kernels)" % pe)
    for f in sorted(glob.glob(os.path.join(out, "pe%d_pmc*" % pe, "**", "*counter_collection.csv"), recursive=True)):
        acc = defaultdict(list)
        for row in csv.DictReader(open(f)):
            if "mk_scan" in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, v in sorted(acc.items()):
            print("  {:28s} n={:3d} avg={:.6g}".format(k, len(v), sum(v) / len(v)))
PY

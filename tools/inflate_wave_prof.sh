#!/bin/bash
# Runs on the GPU box: kernel trace + PMC passes (wave / instruction / wait counters) of the wave-per-member inflate kernel on 64 MB of
# zlib level-6 members.   output: gpurun_out/prof_inflate_wave/summary.txt
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_inflate_wave
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 $ROOT/tools/inflate_wave_prof.py > $OUT/kt.log 2>&1
i=0
for SET in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
  "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_IFETCH SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/pmc$i -o pmc -- python3 $ROOT/tools/inflate_wave_prof.py > $OUT/pmc$i.log 2>&1
  echo "pmc$i rc=$?"
done
python3 - <<PY > $OUT/summary.txt
import csv, glob, re
from collections import defaultdict
print("# tools/inflate_wave_prof.sh: mk_bgzf_inflate_wave_kernel<4096> on 64 MB of zlib level-6 members of BAM records (1 029 members, 3 launches per pass)")
for ln in open("$OUT/kt.log"):
    if "kernel ms" in ln or "bytes of text" in ln: print("# " + ln.rstrip())
for f in glob.glob("$OUT/kt/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "inflate" in r["Name"] or "crc_check" in r["Name"]:
            print("  {:90s} {:>5s} {:>13s} {:>12s}".format(re.sub(r"\(.*", "", r["Name"])[:90], r["Calls"], r["TotalDurationNs"], r["AverageNs"]))
acc = defaultdict(float); cnt = defaultdict(int)
for f in glob.glob("$OUT/pmc*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "inflate_wave" not in row["Kernel_Name"]: continue
        acc[row["Counter_Name"]] += float(row["Counter_Value"]); cnt[row["Counter_Name"]] += 1
print("# PMC, per launch (average over the dispatches of the kernel)")
for k in sorted(acc): print("  %-24s %14.0f  (n=%d)" % (k, acc[k] / max(1, cnt[k]) , cnt[k]))
PY
cat $OUT/summary.txt

#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace of tools/window_prof.py (one FASTQ / BAM / FASTA / paired window each, 4 calls each)
# output: gpurun_out/prof_windows/summary.txt
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_windows
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 $ROOT/tools/window_prof.py > $OUT/kt.log 2>&1
echo "kernel-trace rc=$?"
python3 - <<PY > $OUT/summary.txt
import csv, glob, re
print("# tools/window_prof.sh: rocprofv3 --kernel-trace --stats -- python3 tools/window_prof.py (4 calls per window kind: 1 warm-up + 3)")
for ln in open("$OUT/kt.log"):
    if ln.startswith("{"): print("# " + ln.rstrip()[:1200])
rows = []
for f in glob.glob("$OUT/kt/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -int(r["TotalDurationNs"]))
print("# kernel, calls, total ns, average ns, min ns, max ns")
for r in rows[:48]:
    print("  {:104s} {:>5s} {:>13s} {:>12s} {:>12s} {:>12s}".format(re.sub(r"\(.*", "", r["Name"])[:104], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"]))
PY
cat $OUT/summary.txt

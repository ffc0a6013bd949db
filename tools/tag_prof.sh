#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace of `merkurio tag` BAM -> BAM with the records resident on the device
# (cli/tag_windows.cpp, mk_tag_bam_window) on the 8 M-record BAM of tools/e2e_tag.py, and of the same command with --host-ingest.
# usage: tools/tag_prof.sh [records, default 8000000]     output: gpurun_out/prof_tag/summary.txt
set -u
N=${1:-8000000}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_tag
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/tools/e2e_tag.py $N > $OUT/e2e.txt 2>&1
BIN=$ROOT/merkurio_amd/lib/merkurio
for MODE in resident host; do
  EXTRA=""
  [ $MODE = host ] && EXTRA="--host-ingest"
  MERKURIO_SLOW_EXIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$MODE -o kt -- $BIN tag -f /tmp/e2e_kmers.txt -i /tmp/e2e_in.bam -o /tmp/prof_out_$MODE.bam $EXTRA > $OUT/kt_$MODE.log 2>&1
  echo "kernel-trace $MODE rc=$?"
done
python3 - <<PY > $OUT/summary.txt
import csv, glob, re
print("# tools/tag_prof.sh $N: rocprofv3 --kernel-trace --stats of merkurio tag -f kmers (10 000 31-mers) -i e2e_in.bam ($N x 150-base records) -o out.bam")
for ln in open("$OUT/e2e.txt"):
    if "wall" in ln or "windows on the device" in ln: print("#   " + ln.rstrip())
for mode in ("resident", "host"):
    print()
    print("# %s: kernel trace (ns): name, calls, total, average, min, max" % ("records resident on the device (default)" if mode == "resident" else "--host-ingest (the r04 path)"))
    rows = []
    for f in glob.glob("$OUT/kt_%s/**/*kernel_stats.csv" % mode, recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -int(r["TotalDurationNs"]))
    tot = sum(int(r["TotalDurationNs"]) for r in rows)
    for r in rows[:28]:
        print("  {:100s} {:>5s} {:>13s} {:>12s} {:>12s} {:>12s}".format(re.sub(r"\(.*", "", r["Name"])[:100], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"]))
    print("  all kernels: %.1f ms" % (tot / 1e6))
PY
cat $OUT/summary.txt

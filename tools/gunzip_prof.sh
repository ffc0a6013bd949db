#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel traces of tools/gunzip_bench.py (one-member gzip in parallel pieces) and of
# tools/codec_real.py (BGZF inflate of zlib level-6 members, every kernel variant).
# usage: tools/gunzip_prof.sh [reads, default 4000000] [codec_real MB, default 1024]     output: gpurun_out/prof_gunzip/summary.txt
set -u
READS=${1:-4000000}
MB=${2:-1024}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_gunzip
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/gz -o gz -- python3 $ROOT/tools/gunzip_bench.py $READS 6 > $OUT/gz.log 2>&1
echo "gunzip kernel-trace rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/real -o real -- python3 $ROOT/tools/codec_real.py $MB > $OUT/real.log 2>&1
echo "codec_real kernel-trace rc=$?"
python3 - <<PY > $OUT/summary.txt
import csv, glob, re
for tag, cmd in (("gz", "tools/gunzip_bench.py $READS 6 (3 calls with a wave per piece, 3 with a lane per piece)"), ("real", "tools/codec_real.py $MB (3 calls per size and kernel variant, BAM and FASTQ members)")):
    print("# rocprofv3 --kernel-trace --stats -- python3 " + cmd)
    for ln in open("$OUT/%s.log" % tag):
        if ln.startswith(("gzip -", "BAM", "FASTQ", "  ")) and "rocprofv3" not in ln: print(ln.rstrip()[:330])
    print("# kernel trace (ns): name, calls, total, average, min, max")
    for f in glob.glob("$OUT/%s/**/*kernel_stats.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            if "mk_" in r["Name"]:
                print("  {:80s} {:>5s} {:>13s} {:>12s} {:>12s} {:>12s}".format(re.sub(r"\(.*", "", r["Name"])[:80], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"]))
    print()
PY
cat $OUT/summary.txt | tail -40

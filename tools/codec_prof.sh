#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace of tools/codec_bench.py (deflate + inflate of BAM-shaped
# records through the C ABI) and two PMC passes of their own (wave / instruction / wait counters, LDS conflicts).
# usage: tools/codec_prof.sh [megabytes, default 2048]     output: gpurun_out/prof_codec/summary.txt
set -u
MB=${1:-2048}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_codec
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 $ROOT/tools/codec_bench.py $MB 3 > $OUT/kt.log 2>&1
echo "kernel-trace rc=$?"
i=0
for SET in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/pmc$i -o pmc -- python3 $ROOT/tools/codec_bench.py $MB 1 > $OUT/pmc$i.log 2>&1
  echo "pmc$i [$SET] rc=$?"
done
python3 - <<PY > $OUT/summary.txt
import csv, glob, re
from collections import defaultdict
print("# tools/codec_prof.sh $MB: rocprofv3 --kernel-trace --stats of tools/codec_bench.py $MB 3 (two corpora x 3 deflate calls + 3 + 3 inflate calls), then PMC passes of one call each")
for ln in open("$OUT/kt.log"):
    if ln.startswith(("BAM", "  deflate", "  inflate")): print(ln.rstrip())
print()
print("# kernel trace (ns): name, calls, total, average, min, max")
for f in glob.glob("$OUT/kt/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mk_bgzf" in r["Name"] or "copyBuffer" in r["Name"]:
            print("  {:95s} {:>5s} {:>13s} {:>12s} {:>12s} {:>12s}".format(re.sub(r"\(.*", "", r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"]))
print()
print("# PMC (summed over the dispatches of each kernel in a pass of one deflate call per corpus, one inflate call per member set)")
acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(int)); res = {}
for f in glob.glob("$OUT/pmc*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", row["Kernel_Name"])
        if "mk_bgzf" not in k: continue
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[k][row["Counter_Name"]] += 1
        res[k] = (row.get("VGPR_Count"), row.get("LDS_Block_Size"), row.get("Scratch_Size"), row.get("Workgroup_Size"))
for k in sorted(acc):
    print("  {}  VGPR={} LDS={} scratch={} workgroup={}".format(k, *res[k]))
    c = acc[k]
    for name in sorted(c): print("      {:22s} {:.6g}  ({} dispatches)".format(name, c[name], cnt[k][name]))
    if c.get("SQ_WAVE_CYCLES"):
        print("      -> waiting {:.1%} of the wave cycles, issuing {:.1%}; VALU {:.1%}, LDS {:.1%}".format(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"], c["SQ_ACTIVE_INST_LDS"] / c["SQ_WAVE_CYCLES"]))
    if c.get("SQ_LDS_IDX_ACTIVE"):
        print("      -> LDS bank conflict cycles / LDS active cycles = {:.2f}".format(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]))
PY
cat $OUT/summary.txt

#!/usr/bin/env python3
"""Summarises a tools/profile_gpu.sh output directory: per-kernel stats from the kernel trace
and per-launch averages of every PMC counter for the scan kernel."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


for f in find("kt/**/*kernel_stats.csv"):
    print("== kernel stats:", os.path.relpath(f, out))
    for row in csv.DictReader(open(f)):
        print("  {Name:60.60s} calls={Calls} avg_ns={AverageNs} total_ns={TotalDurationNs} pct={Percentage}".format(**row))

for f in find("kt/**/*kernel_trace.csv"):
    d = defaultdict(list)
    for row in csv.DictReader(open(f)):
        d[row["Kernel_Name"]].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        last = row
    print("== kernel trace:", os.path.relpath(f, out))
    for k, v in d.items():
        print("  {:60.60s} n={} avg_us={:.1f} min_us={:.1f} max_us={:.1f}".format(k, len(v), sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3))
    for key in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size"):
        if key in last:
            print("  last launch {}={}".format(key, last[key]))

print("== PMC (per-launch average over launches of kernels matching 'mk_scan')")
for f in find("pmc*/**/*counter_collection.csv"):
    acc = defaultdict(list)
    for row in csv.DictReader(open(f)):
        if "mk_scan" not in row["Kernel_Name"]:
            continue
        acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print("  {:28s} n={:3d} avg={:.6g}".format(k, len(v), sum(v) / len(v)))

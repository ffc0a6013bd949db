#!/usr/bin/env python3
"""Summarises a tools/profile_gpu.sh output directory: per-kernel stats from the kernel trace,
per-launch averages of every PMC counter for the scan kernel, and traffic.json -- the HBM bytes per
launch (FETCH_SIZE / WRITE_SIZE passes, gfx950 correction) stamped with the hash of the kernel
source it was measured on (bench.py refuses a figure from another kernel body)."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

out = sys.argv[1]
bench_args = sys.argv[2] if len(sys.argv) > 2 else ""
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


for f in find("kt/**/*kernel_stats.csv") + find("kt*kernel_stats.csv"):
    print("== kernel stats:", os.path.relpath(f, out))
    shutil.copy(f, os.path.join(out, "kernel_stats.csv"))
    for row in csv.DictReader(open(f)):
        print("  {Name:60.60s} calls={Calls} avg_ns={AverageNs} min_ns={MinNs} total_ns={TotalDurationNs} pct={Percentage}".format(**row))

scan_name, scan_avg_us = None, None
for f in find("kt/**/*kernel_trace.csv") + find("kt*kernel_trace.csv"):
    d = defaultdict(list)
    last = None
    for row in csv.DictReader(open(f)):
        d[row["Kernel_Name"]].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        last = row
    print("== kernel trace:", os.path.relpath(f, out))
    for k, v in d.items():
        print("  {:60.60s} n={} avg_us={:.1f} min_us={:.1f} max_us={:.1f}".format(k, len(v), sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3))
        if "mk_scan_kernel" in k:
            scan_name, scan_avg_us = k, sum(v) / len(v) / 1e3
    for key in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size"):
        if last and key in last:
            print("  last launch {}={}".format(key, last[key]))

print("== PMC (per-launch average over launches of kernels matching 'mk_scan')")
pmc = {}
for f in find("pmc*/**/*counter_collection.csv"):
    acc = defaultdict(list)
    for row in csv.DictReader(open(f)):
        if "mk_scan" not in row["Kernel_Name"]:
            continue
        acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in sorted(acc.items()):
        pmc[k] = sum(v) / len(v)
        print("  {:28s} n={:3d} avg={:.6g}".format(k, len(v), sum(v) / len(v)))

if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc and scan_name:
    import re
    import bench
    a = bench.parse_args(bench_args.split())
    short = re.sub(r"^void mk::", "", scan_name).split("(")[0].replace(" ", "")
    # bench.py's kernel names leave the stream-load flavour (5th template argument) out / call it "plain"
    m5 = re.match(r"(mk_scan_kernel<[^>]*),(true|false|0|1)>$", short)
    if m5:
        short = m5.group(1) + (">" if m5.group(2) in ("true", "1") else ",plain>")
    fetch_kb, write_kb = pmc["FETCH_SIZE"], pmc["WRITE_SIZE"]
    j = {
        "source": "tools/profile_gpu.sh: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, --kernel-trace only",
        "command": "python3 bench.py " + bench_args,
        "kernel": short,
        "kernel_source_sha16": bench.kernel_source_hash(),
        "records_per_gpu": a.records, "read_len": a.read_len, "patterns": a.patterns * (2 if a.rc else 1),
        "FETCH_SIZE_KB_per_launch": fetch_kb, "WRITE_SIZE_KB_per_launch": write_kb,
        "correction": "FETCH_SIZE x2 on gfx950 (16 B/lane streaming reads are tallied at half size, MI355X_MICROARCH.md HBM section); WRITE_SIZE exact",
        "hbm_bytes_per_launch": fetch_kb * 1024 * 2 + write_kb * 1024,
        "kernel_avg_us_rocprof": scan_avg_us,
    }
    json.dump(j, open(os.path.join(out, "traffic.json"), "w"), indent=1)
    print("== traffic.json", json.dumps(j))

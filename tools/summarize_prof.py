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
    scan_row = None  # a launch of the scan kernel: ITS resources are the ones worth printing (r03 printed the last launch's,
    for row in csv.DictReader(open(f)):  # i.e. those of whatever tiny kernel ran last)
        d[row["Kernel_Name"]].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        if "mk_scan_kernel" in row["Kernel_Name"]:
            scan_row = row
    print("== kernel trace:", os.path.relpath(f, out))
    for k, v in d.items():
        print("  {:60.60s} n={} avg_us={:.1f} min_us={:.1f} max_us={:.1f}".format(k, len(v), sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3))
        # the scan kernel of the run = the mk_scan_kernel variant launched most often (a hit-dense run launches the
        # sparse flavour during its warm-up and the dense one afterwards)
        if "mk_scan_kernel" in k and (scan_name is None or len(v) > len(d[scan_name])):
            scan_name, scan_avg_us = k, sum(v) / len(v) / 1e3
    for key in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size"):
        if scan_row and key in scan_row:
            print("  scan kernel launch {}={}  (as the trace reports it; the compiler's own figures: profiles/r04_isa_resources.txt)".format(key, scan_row[key]))

print("== PMC (per-launch average over the launches of %s)" % (scan_name or "kernels matching 'mk_scan'"))
pmc = {}
for f in find("pmc*/**/*counter_collection.csv"):
    acc = defaultdict(list)
    for row in csv.DictReader(open(f)):
        if "mk_scan" not in row["Kernel_Name"] or (scan_name and row["Kernel_Name"] != scan_name):
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
    # bench.py's kernel names (scan_kernel.hip: launch_scan) from the template arguments <S, QC, EMIT, GF, FL, MC>
    targs = re.match(r"mk_scan_kernel<(.*)>$", short)
    if targs:
        t = targs.group(1).split(",")
        fl = t[4] if len(t) > 4 else "1"
        mc = t[5] if len(t) > 5 else "0"
        short = "mk_scan_kernel<" + ",".join(t[:4]) + (",plain" if fl in ("0", "false") and mc in ("0", "false") else "")
        if mc not in ("0", "false"):
            short += (",s2=%s" % mc if mc not in ("1", "true") else "") + ",2-class"
        short += ">"
    fetch_kb, write_kb = pmc["FETCH_SIZE"], pmc["WRITE_SIZE"]
    # gfx950 tallies a wide coalesced streaming read at half its bytes (MI355X_MICROARCH.md, HBM): that correction
    # applies to the kernel's TEXT STREAM only (16 B per lane, every byte once + one halo chunk per 31-chunk tile),
    # not to its random 8- / 32-byte reads of filter blocks, table buckets and occurrence windows.  The stream's
    # bytes are known, so: hbm = FETCH_SIZE + stream / 2 + WRITE_SIZE (r03 doubled all of FETCH_SIZE, which inflated
    # the global-filter configuration, whose fetches are mostly random, from ~6.0 to 8.9 GB).
    n_text = a.records * a.read_len
    stream = n_text * 32 // 31
    fetch_b = fetch_kb * 1024
    stream_reported = min(fetch_b, stream / 2)
    j = {
        "source": "tools/profile_gpu.sh: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, --kernel-trace only",
        "command": "python3 bench.py " + bench_args,
        "kernel": short,
        "kernel_source_sha16": bench.kernel_source_hash(),
        "records_per_gpu": a.records, "read_len": a.read_len, "patterns": a.patterns * (2 if a.rc else 1), "plant_every": a.plant_every,
        "FETCH_SIZE_KB_per_launch": fetch_kb, "WRITE_SIZE_KB_per_launch": write_kb,
        "correction": "gfx950 tallies 16 B/lane streaming reads at half size (MI355X_MICROARCH.md HBM section): the text stream "
                      "(records x read_len x 32/31 bytes, known) is counted twice, the random reads (the rest of FETCH_SIZE) once; WRITE_SIZE exact",
        "text_stream_bytes": stream,
        "random_fetch_bytes_per_launch": fetch_b - stream_reported,
        "hbm_bytes_per_launch": fetch_b + stream_reported + write_kb * 1024,
        "hbm_bytes_per_launch_all_doubled_r03": fetch_b * 2 + write_kb * 1024,
        "raw_requests": {k: pmc[k] for k in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_BUBBLE_sum", "TCC_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum") if k in pmc},
        "kernel_avg_us_rocprof": scan_avg_us,
    }
    json.dump(j, open(os.path.join(out, "traffic.json"), "w"), indent=1)
    print("== traffic.json", json.dumps(j))

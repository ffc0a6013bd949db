#!/usr/bin/env python3
"""Throughput and compression ratio of the device BGZF codec (mk_bgzf_deflate / mk_bgzf_inflate through the C ABI) on
BAM-shaped records, next to zlib on one host thread.   usage: tools/codec_bench.py [megabytes, default 512] [repeats]"""
import os, sys, time, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from merkurio_amd import native as mk
from textio import bam_like
from test_gpu_codec import zlib_bgzf

mb = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
codec = mk.Codec()
for label, const_qual in (("BAM records, 4-bin qualities", False), ("BAM records, constant qualities", True)):
    unit = bam_like(200000, seed=21, const_qual=const_qual)
    n_unit = max(1, mb * (1 << 20) // len(unit))
    data = unit * n_unit  # (repeats lie 55 MB apart: outside DEFLATE's 32 KiB window)
    sample = data[:32 * 65280]
    z = {}
    for lvl in (1, 6):
        t0 = time.time()
        z[lvl] = len(zlib_bgzf(sample, level=lvl))
        z[lvl] = (z[lvl], len(sample) / (time.time() - t0) / 1e6)
    print(f"{label}: {len(data) / 1e6:.0f} MB; zlib on one thread (2 MB sample): level 1 ratio {len(sample) / z[1][0]:.2f} at {z[1][1]:.0f} MB/s, "
          f"level 6 ratio {len(sample) / z[6][0]:.2f} at {z[6][1]:.0f} MB/s", flush=True)
    for r in range(reps):
        t0 = time.time()
        blob = codec.deflate(data)
        dt = time.time() - t0
        up, dev, down = codec.times()
        print(f"  deflate: {dt * 1e3:.0f} ms wall ({len(data) / dt / 1e9:.2f} GB/s), upload {up:.0f} ms, kernels {dev:.1f} ms "
              f"({len(data) / dev / 1e6:.1f} GB/s), download {down:.0f} ms; ratio {len(data) / len(blob):.2f}", flush=True)
    for r in range(reps):
        t0 = time.time()
        text = codec.inflate(blob)
        dt = time.time() - t0
        up, dev, down = codec.times()
        print(f"  inflate (device-written members): {dt * 1e3:.0f} ms wall ({len(data) / dt / 1e9:.2f} GB/s), upload {up:.0f} ms, kernels {dev:.1f} ms "
              f"({len(data) / dev / 1e6:.1f} GB/s), download {down:.0f} ms", flush=True)
    assert text == data
    zb = zlib_bgzf(data[:64 << 20], level=6)
    for r in range(reps):
        t0 = time.time()
        text = codec.inflate(zb)
        dt = time.time() - t0
        up, dev, down = codec.times()
        print(f"  inflate (zlib level 6 members, 64 MB of text): {dt * 1e3:.0f} ms wall, kernels {dev:.1f} ms ({len(text) / dev / 1e6:.1f} GB/s)", flush=True)
    assert text == data[:64 << 20]

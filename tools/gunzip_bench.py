#!/usr/bin/env python3
"""One-member gzip inflated in parallel pieces on the device (mk_gzip_inflate_device) against zlib on one host thread.
usage: tools/gunzip_bench.py [reads, default 4 000 000 = 1.27 GB of FASTQ] [gzip level, default 1 and 6]"""
import os, sys, time, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from merkurio_amd import native as mk
sys.argv = sys.argv[:1] + sys.argv[1:]
from bench import _fastq_binned
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
levels = [int(x) for x in sys.argv[2:] if not x.startswith("--")] or [1, 6]
chunks = []
for a in sys.argv[2:]:
    if a.startswith("--chunks="):
        chunks = [int(k) for k in a[9:].split(",")]
data = _fastq_binned(n)
codec = mk.Codec()
for level in levels:
    t0 = time.time()
    co = zlib.compressobj(level, zlib.DEFLATED, 31)
    gz = co.compress(data) + co.flush()
    t_c = time.time() - t0
    t0 = time.time()
    ref = zlib.decompress(gz, 31)
    t_z = time.time() - t0
    assert ref == data
    del ref
    for which, chunk, label in [(0, 0, "a wave per piece, cuts by the stream's size")] + [(0, k << 10, f"a wave per piece, cuts every {k} KiB") for k in chunks] + [(1, 0, "a lane per piece")]:
        codec.set_inflate_kernel(which)
        codec.set_gzip_chunk(chunk)
        best = None
        for rep in range(3):
            text = codec.gunzip(gz)
            assert text is not None, codec.gzip_info
            if best is None or codec.last_call_s < best[0]:
                best = (codec.last_call_s, codec.last_read_s, codec.gzip_info)
            assert text == data
            del text
        seg, ms = best[2]
        print(f"gzip -{level}, {label}: {len(data) / 1e6:.0f} MB of FASTQ in {len(gz) / 1e6:.0f} MB (ratio {len(data) / len(gz):.2f}; written in {t_c:.0f} s); zlib inflate on one thread "
              f"{t_z:.2f} s = {len(data) / t_z / 1e9:.2f} GB/s; device: {best[0] * 1e3:.0f} ms = {len(data) / best[0] / 1e9:.1f} GB/s of text in {seg} pieces "
              f"(upload {ms[0]}, block search {ms[1]}, pieces {ms[2]}, resolution {ms[3]}, CRC {ms[4]} ms); + text to the host {best[1] * 1e3:.0f} ms", flush=True)

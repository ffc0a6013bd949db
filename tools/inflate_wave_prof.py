#!/usr/bin/env python3
"""64 MB of zlib level-6 BGZF members of BAM records through mk_bgzf_inflate (the wave-per-member kernel), for rocprofv3 (kernel trace
or one --pmc set per run).   usage: rocprofv3 ... -- python3 tools/inflate_wave_prof.py [megabytes, default 64]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from textio import bam_like
from merkurio_amd import native as mk
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
unit = bam_like(200000, seed=21)
raw = (unit * ((mb << 20) // len(unit) + 1))[:mb << 20]
zb = bench._zlib_bgzf(raw)
tokens = 0
codec = mk.Codec()
for _ in range(3):
    assert codec.inflate(zb) == raw
    print("kernel ms", codec.times()[1], flush=True)
print(f"{len(raw)} bytes of text, {len(zb)} bytes of BGZF, {len(raw) // 65280 + 1} members")

#!/usr/bin/env python3
"""Kernel time of mk_bgzf_deflate by the number of members in the call (BAM-shaped records): where the resident grid of 16 waves per
CU (4 096 members at once on an MI355X) leaves a tail.   usage: tools/deflate_members_sweep.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from textio import bam_like
from merkurio_amd import native as mk
codec = mk.Codec()
unit = bam_like(200000, seed=21)
text = unit * (9000 * 65280 // len(unit) + 1)
for members in (1000, 2000, 3000, 3500, 4000, 4096, 4200, 4500, 5000, 6000, 8192, 8300):
    part = text[:members * 65280]
    best = None
    for _ in range(3):
        blob = codec.deflate(part)
        up, dev, down = codec.times()
        best = dev if best is None else min(best, dev)
    print(f"{members:5d} members ({len(part) / 1e6:6.0f} MB): kernels {best:6.2f} ms = {len(part) / best / 1e6:5.1f} GB/s of text", flush=True)

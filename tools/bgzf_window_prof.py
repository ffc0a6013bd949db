#!/usr/bin/env python3
"""One bgzip'ed FASTQ window through mk_extract_fastq_bgzf and mk_extract_fastq_text (bench.py: bgzf_window_config), for
rocprofv3 --kernel-trace: which kernels a window costs.   usage: rocprofv3 --kernel-trace --stats ... -- python3 tools/bgzf_window_prof.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from merkurio_amd import native as mk
codec = mk.Codec()
print(json.dumps(bench.bgzf_window_config(mk, codec, 3)))

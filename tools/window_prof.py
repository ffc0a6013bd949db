#!/usr/bin/env python3
"""Which kernels a window costs (for rocprofv3 --kernel-trace --stats): one bgzip'ed FASTQ window through mk_extract_fastq_bgzf /
mk_extract_fastq_text (bench.py: bgzf_window_config), one BAM window through mk_tag_bam_window (bam_window_config), one FASTA
window (wrapped at 60 columns, 1 000 records of 0.5 Mbp) and one paired FASTQ window through mk_extract_window.
usage: rocprofv3 --kernel-trace --stats ... -- python3 tools/window_prof.py"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from merkurio_amd import native as mk
codec = mk.Codec()
print(json.dumps(bench.bgzf_window_config(mk, codec, 3)))
print(json.dumps(bench.bam_window_config(mk, codec, 3)))
rng = np.random.default_rng(4)
pats = [bytes(x) for x in np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(10_000, 31))]]
m = mk.Matcher(mk.parse_pattern_list(kmer_seq=pats), device=0)
# FASTA: 1 000 records x 0.5 Mbp at 60 columns = 508 MB of text
W, rec_bases, n_fa = 60, 500_040, 1000
lines = rec_bases // W
body = np.empty((n_fa, lines, W + 1), dtype=np.uint8)
body[:, :, :W] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n_fa, lines, W))]
body[:, :, W] = 10
fa = b"".join(b">chr%d some description\n" % i + body[i].tobytes() for i in range(n_fa))
del body
for rep in range(4):
    t0 = time.perf_counter()
    r = m.extract_window([{"text": fa, "ends_at_record": True}], fmt=mk.MK_TEXT_FASTA, logging=False, want=("tail",))
    dt = time.perf_counter() - t0
assert r["status"] == 0 and r["n_rec"] == n_fa
print(json.dumps({"workload": f"FASTA window: {n_fa} records x {rec_bases} bases at {W} columns = {len(fa) / 1e6:.0f} MB of text, 10 000 31-mers, any-hit flags",
                  "ms_per_call_incl_python": round(dt * 1e3, 1), "records_kept": int(sum(r["keep"]))}))
del fa
# paired FASTQ: 2 x 1.5 M reads of 150 bases
n, L, H = 1_500_000, 150, 13
def fastq(seed):
    g = np.random.default_rng(seed)
    rec = np.empty((n, H + L + 3 + L + 1), dtype=np.uint8)
    rec[:, :H] = np.array([b"@r%010d\n" % i for i in range(n)], dtype="S13").view(np.uint8).reshape(n, H)
    rec[:, H:H + L] = np.frombuffer(b"ACGT", dtype=np.uint8)[g.integers(0, 4, size=(n, L))]
    rec[:, H + L:H + L + 3] = np.frombuffer(b"\n+\n", dtype=np.uint8)
    rec[:, H + L + 3:H + 2 * L + 3] = ord("I")
    rec[:, -1] = 10
    return rec.tobytes()
f1, f2 = fastq(1), fastq(2)
for rep in range(4):
    t0 = time.perf_counter()
    r = m.extract_window([{"text": f1, "ends_at_record": True}, {"text": f2, "ends_at_record": True}], fmt=mk.MK_TEXT_FASTQ, logging=False, want=("tail",))
    dt = time.perf_counter() - t0
assert r["status"] == 0 and r["n_rec"] == n
print(json.dumps({"workload": f"paired FASTQ window: 2 x {n} x {L} bp = {2 * len(f1) / 1e6:.0f} MB of text, 10 000 31-mers, any-hit flags", "ms_per_call_incl_python": round(dt * 1e3, 1)}))

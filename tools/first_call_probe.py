#!/usr/bin/env python3
"""Where the first calls of a process spend their time (round 4: end-to-end runs are start-up bound): runtime start, the
first matcher, the first and second text window through mk_extract_fastq_text (code-object loads, first allocations)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MERKURIO_SYSTEM_HIP", "1")
import numpy as np
t0 = time.perf_counter()
from merkurio_amd import native as mk
L = mk.load()
t1 = time.perf_counter()
n = L.mk_device_count()
t2 = time.perf_counter()
rng = np.random.default_rng(1)
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
pats = [acgt[rng.integers(0, 4, 31)].tobytes() for _ in range(10000)]
patterns = mk.parse_pattern_list(kmer_seq=pats)
t3 = time.perf_counter()
m = mk.Matcher(patterns)
t4 = time.perf_counter()
m2 = mk.Matcher(patterns)
t5 = time.perf_counter()
nrec = 100_000
text = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, acgt[rng.integers(0, 4, 150)].tobytes(), b"I" * 150) for i in range(nrec))
ts = []
for k in range(3):
    a = time.perf_counter()
    r = m.extract_fastq_text(text, logging=False)
    ts.append(time.perf_counter() - a)
print("import + dlopen %.3f s | mk_device_count (runtime start) %.3f s | first matcher %.3f s | second matcher %.3f s" % (t1 - t0, t2 - t1, t4 - t3, t5 - t4))
print("mk_extract_fastq_text on a 32 MB window: first call %.3f s, second %.3f s, third %.3f s" % tuple(ts))

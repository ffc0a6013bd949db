#!/usr/bin/env python3
"""Times the one-off pattern-set compilation at the BASELINE sizes:
pattern-list construction (sort / dedup / reverse complements) on the host and
mk_matcher_create (filter + exact table build + upload)."""
import random
import sys
import time

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from merkurio_amd import native as mk  # noqa: E402

for n, k, rc in ((1024, 31, True), (10_000, 31, False), (500_000, 21, False), (500_000, 21, True)):
    rng = random.Random(n + k)
    kmers = ["".join(rng.choice("ACGT") for _ in range(k)) for _ in range(n)]
    t0 = time.perf_counter()
    patterns = mk.parse_pattern_list(kmer_seq=kmers, reverse_complement=rc)
    t1 = time.perf_counter()
    m = mk.Matcher(patterns)
    t2 = time.perf_counter()
    print("%7d %d-mers rc=%-5s -> %7d patterns: list %.3f s, matcher_create %.3f s  %s %s" % (
        n, k, rc, len(patterns), t1 - t0, t2 - t1, m.filter_info(), m.filter_mode()), flush=True)

#!/usr/bin/env python3
"""Times the one-off pattern-set compilation at the BASELINE sizes: mk_parse_pattern_list
(case / reverse complement / canonical, sort, dedup -- all host threads) and mk_matcher_create (length classes,
upload, filter images + exact table built on the device).  The C calls are timed on pre-packed arrays, as the C++
host program makes them; the first matcher of a process also pays the HIP runtime's start-up, shown separately."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from merkurio_amd import native as mk  # noqa: E402

L = mk.load()
t0 = time.perf_counter()
mk.Matcher([b"ACGTACGTACGTACGTACGTACGTACGTACG"]).close()
print("first matcher of the process (HIP runtime start-up + 1 pattern): %.3f s" % (time.perf_counter() - t0), flush=True)
rng = np.random.default_rng(1)
print("host threads: %d" % len(os.sched_getaffinity(0)))
for n, k, rc in ((1024, 31, 1), (10_000, 31, 0), (500_000, 21, 0), (500_000, 21, 1), (1_000_000, 21, 0)):
    data = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n, k), dtype=np.uint8)].reshape(-1).copy()
    off = np.arange(n + 1, dtype=np.uint32) * k
    best = [9e9, 9e9]
    for rep in range(3):
        ob, oo, on = C.c_void_p(), C.c_void_p(), C.c_uint32()
        t0 = time.perf_counter()
        mk._check(L.mk_parse_pattern_list(data.ctypes.data, off.ctypes.data, n, rc, 0, 0, 0, C.byref(ob), C.byref(oo), C.byref(on)))
        t1 = time.perf_counter()
        h = C.c_void_p()
        mk._check(L.mk_matcher_create(ob, oo, on.value, mk.MK_ALGO_AUTO, 0, 0, 0, C.byref(h)))
        t2 = time.perf_counter()
        q, s, e, tb = C.c_uint32(), C.c_uint32(), C.c_uint64(), C.c_uint64()
        L.mk_matcher_filter_info(h, C.byref(q), C.byref(s), C.byref(e), C.byref(tb))
        L.mk_matcher_destroy(h)
        L.mk_free(ob)
        L.mk_free(oo)
        best = [min(best[0], t1 - t0), min(best[1], t2 - t1)]
    print("%8d %d-mers rc=%d -> %8d patterns: list %.4f s, matcher_create %.4f s, together %.4f s  (q=%d S=%d, %d entries, table %d MiB)" % (
        n, k, rc, on.value, best[0], best[1], best[0] + best[1], q.value, s.value, e.value, tb.value >> 20), flush=True)

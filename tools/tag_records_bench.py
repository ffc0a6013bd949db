#!/usr/bin/env python3
"""mk_tag_records / mk_extract_single on a batch in which EVERY record carries a k-mer (tag on already extracted
reads): where the call spends its time -- upload, device (scan + emission order + rows + counts + per-record pattern
sets, all kernels), download, host loops (mk_matcher_batch_times).  usage: tools/tag_records_bench.py [n_records]"""
import ctypes as C
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from merkurio_amd import native as mk

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
L = 150
rng = np.random.default_rng(5)
pats = [bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 31)]) for _ in range(10000)]
patterns = mk.parse_pattern_list(kmer_seq=pats)
lib = mk.load()
for algo, name in ((mk.MK_ALGO_AC, "Aho-Corasick"),):
    m = mk.Matcher(patterns, algo=algo)
    seq = np.zeros(n * L, dtype=np.uint8)
    off = np.zeros(n + 1, dtype=np.uint64)
    assert lib.mk_synth_reads_host(m.handle, 99, 0, n, L, 1, seq.ctypes.data, off.ctypes.data) == 0
    keep = np.zeros(n, dtype=np.uint8)
    foff = np.zeros(n + 1, dtype=np.uint64)
    fpat = np.zeros(2 * n, dtype=np.uint32)
    rows = np.zeros(2 * n, dtype=mk.ROW_DTYPE)
    for logging in (0, 1):
        for rep in range(3):
            c, counts, n_rows = mk.Counters(), np.zeros(len(patterns), dtype=np.uint32), C.c_uint64()
            t0 = time.perf_counter()
            rc = lib.mk_tag_records(m.handle, seq.ctypes.data, off.ctypes.data, n, logging, 1, 0, keep.ctypes.data, rows.ctypes.data, len(rows),
                                    C.byref(n_rows), C.byref(c), counts.ctypes.data, foff.ctypes.data, fpat.ctypes.data, len(fpat))
            dt = time.perf_counter() - t0
            assert rc == 0, lib.mk_last_error()
        t = m.batch_times_ms()
        print(f"mk_tag_records, {name}, {n} records x {L} bp, every record hits, logging={logging}: {dt * 1e3:.1f} ms wall = upload {t['upload']:.1f} + "
              f"device {t['device']:.1f} + download {t['download']:.1f} + host {t['host']:.1f} ms; {int(foff[n])} set entries, {n_rows.value} rows, "
              f"kept {int(keep.sum())}", flush=True)
    for rep in range(3):
        c, counts, n_rows = mk.Counters(), np.zeros(len(patterns), dtype=np.uint32), C.c_uint64()
        t0 = time.perf_counter()
        rc = lib.mk_extract_single(m.handle, seq.ctypes.data, off.ctypes.data, n, 1, 0, keep.ctypes.data, rows.ctypes.data, len(rows),
                                   C.byref(n_rows), C.byref(c), counts.ctypes.data)
        dt = time.perf_counter() - t0
        assert rc == 0, lib.mk_last_error()
    t = m.batch_times_ms()
    print(f"mk_extract_single, {name}, logging=1: {dt * 1e3:.1f} ms wall = upload {t['upload']:.1f} + device {t['device']:.1f} + download "
          f"{t['download']:.1f} + host {t['host']:.1f} ms; {n_rows.value} rows", flush=True)

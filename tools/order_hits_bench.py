#!/usr/bin/env python3
"""Emission order of a hit-dense batch: the hand-written device ordering (mk_order_hits_device: record bins +
LDS sort per bin, what mk_scan_batch uses) against the host sort (mk_order_hits, one thread) on the same tuples.
usage: tools/order_hits_bench.py [n_reads]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from merkurio_amd import native as mk

n_rec = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
L = 150
rng = np.random.default_rng(5)
pats = [bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 31)]) for _ in range(10000)]
patterns = mk.parse_pattern_list(kmer_seq=pats, reverse_complement=True)
m = mk.Matcher(patterns)
lib = mk.load()
dev = torch.device("cuda", 0)
d_seq = torch.empty(n_rec * L + 64, dtype=torch.uint8, device=dev)
d_off = torch.empty(n_rec + 1, dtype=torch.int64, device=dev)
st = torch.cuda.current_stream().cuda_stream
for plant_every in (1, 10):
    assert lib.mk_synth_reads_device(m.handle, 99, n_rec, L, plant_every, d_seq.data_ptr(), d_off.data_ptr(), st) == 0
    d_flags = torch.zeros(n_rec + 8, dtype=torch.uint8, device=dev)
    cap = n_rec // plant_every + (1 << 16)
    d_hits = torch.zeros(2 * cap, dtype=torch.int64, device=dev)
    d_nh = torch.zeros(1, dtype=torch.int64, device=dev)
    lib.mk_matcher_hint_hit_density(m.handle, 1000 // plant_every)
    assert lib.mk_scan_device(m.handle, d_seq.data_ptr(), n_rec * L, d_off.data_ptr(), n_rec, mk.MK_MODE_HITS, d_flags.data_ptr(),
                              d_hits.data_ptr(), cap, d_nh.data_ptr(), None, st) == 0
    torch.cuda.synchronize()
    nh = int(d_nh.item())
    host = np.frombuffer(d_hits.cpu().numpy().tobytes(), dtype=mk.HIT_DTYPE)[:nh].copy()
    t0 = time.perf_counter()
    lib.mk_order_hits(m.handle, host.ctypes.data, nh)
    t_host = time.perf_counter() - t0
    keep = d_hits.clone()
    lib.mk_order_hits_device(m.handle, d_hits.data_ptr(), nh, st)  # first call: allocates the scratch
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        d_hits.copy_(keep)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        lib.mk_order_hits_device(m.handle, d_hits.data_ptr(), nh, st)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    devs = np.frombuffer(d_hits.cpu().numpy().tobytes(), dtype=mk.HIT_DTYPE)[:nh]
    assert np.array_equal(devs, host)
    info = m.order_info()
    print(f"{n_rec} reads x {L} bp, 1 in {plant_every} hits: {nh} tuples; host sort (1 thread) {t_host * 1e3:.1f} ms, "
          f"device order {min(ts) * 1e3:.3f} ms min / {np.median(ts) * 1e3:.3f} ms median (same order: yes; path {info['path']}, "
          f"{info['bins']} bins, largest {info['max_bin']})", flush=True)

#!/usr/bin/env python3
"""order_skew.py -- which path of the device ordering (order_hits.hip) a genome-like batch takes: ONE record of
200 Mbp with a repeat k-mer planted every ~2 kbp plus 2 000 k-mers planted at random (a chromosome with an
interspersed repeat), scanned in hits mode, tuples ordered on the device and compared with the reference's comparator
(numpy).  Prints the path (1 record bins, 2 bins on the whole (record, end, pattern) key, 3 library merge sort), the
bins and the time.  Used by tests/test_gpu_order.py::test_order_on_a_genome_like_batch and profiles/r04_order_skew.txt."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(mk, torch, n_bytes=200_000_000, repeat_every=2000, n_pat=2000):
    import numpy as np
    lib = mk.load()
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(42)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    pats = [acgt[rng.integers(0, 4, 31)].tobytes() for _ in range(n_pat)]
    patterns = mk.parse_pattern_list(kmer_seq=pats)
    rep = np.frombuffer(patterns[0], dtype=np.uint8)
    seq = acgt[rng.integers(0, 4, n_bytes)]
    for at in range(1000, n_bytes - 31, repeat_every):
        at += int(rng.integers(0, 500))
        seq[at:at + 31] = rep
    for i, at in enumerate(rng.integers(0, n_bytes - 31, 50_000).tolist()):
        seq[at:at + 31] = np.frombuffer(patterns[1 + i % (len(patterns) - 1)], dtype=np.uint8)
    m = mk.Matcher(patterns)
    d_seq = torch.zeros(n_bytes + 64, dtype=torch.uint8, device=dev)
    d_seq[:n_bytes] = torch.from_numpy(seq).to(dev)
    d_off = torch.tensor([0, n_bytes], dtype=torch.int64, device=dev)
    d_flags = torch.zeros(8, dtype=torch.uint8, device=dev)
    cap = 1 << 20
    d_hits = torch.zeros(2 * cap, dtype=torch.int64, device=dev)
    d_nh = torch.zeros(1, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    mk._check(lib.mk_scan_device(m.handle, d_seq.data_ptr(), n_bytes, d_off.data_ptr(), 1, mk.MK_MODE_HITS, d_flags.data_ptr(),
                                 d_hits.data_ptr(), cap, d_nh.data_ptr(), None, st))
    torch.cuda.synchronize()
    n = int(d_nh.item())
    assert n <= cap
    h = np.frombuffer(d_hits[:2 * n].cpu().numpy().tobytes(), dtype=mk.HIT_DTYPE).copy()
    t0 = time.perf_counter()
    mk._check(lib.mk_order_hits_device(m.handle, d_hits.data_ptr(), n, st))
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    got = np.frombuffer(d_hits[:2 * n].cpu().numpy().tobytes(), dtype=mk.HIT_DTYPE)
    order = np.lexsort((h["pat"].astype(np.int64), h["pos"].astype(np.int64), h["rec"]))  # equal lengths: end order == start order
    return {"workload": f"one record of {n_bytes} bp, a repeat 31-mer every ~{repeat_every} bp + 50 000 random plants of {n_pat} 31-mers",
            "tuples": n, "repeat_tuples": int((h["pat"] == 0).sum()), "order": m.order_info(), "order_ms": round(ms, 3),
            "ordered_like_the_reference": bool(np.array_equal(got, h[order]))}


if __name__ == "__main__":
    import torch
    from merkurio_amd import native as mk
    print(json.dumps(run(mk, torch)))

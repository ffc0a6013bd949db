"""world_size-2 gloo test of the record-shard / merge / counter-reduction logic (CPU only).
The scan itself is played by the oracle here (tests may use it); what is under test is the
product's sharding module: contiguous shards, pairs never split, ordered merge == unsharded
result, summed counter vector == unsharded counters."""
import os
import random
import socket

import numpy as np
import pytest

import oracle_binding as ob
from merkurio_amd import sharding
from merkurio_amd.native import HIT_DTYPE


def _workload(seed=5, n_rec=301, n_pat=40):
    rnd = random.Random(seed)
    raw = [bytes(rnd.choice(b"ACGT") for _ in range(21)) for _ in range(n_pat)]
    rc, patterns = ob.parse_pattern_list(raw)
    recs = []
    for i in range(n_rec):
        s = bytearray(rnd.choice(b"ACGT") for _ in range(rnd.choice([80, 150, 151])))
        if i % 3 == 0:
            p = rnd.choice(patterns)
            k = rnd.randrange(0, len(s) - len(p) + 1)
            s[k:k + len(p)] = p
        recs.append(bytes(s))
    return patterns, recs


def _scan(patterns, recs):
    """oracle stand-in for mk_scan_batch: (flags, hits, counter vector)"""
    m = ob.Matcher(patterns, True, 0, False)
    keep, rows, c, found = ob.tag_records(m, recs, logging=True)
    hits = np.zeros(len(rows), dtype=HIT_DTYPE)
    for i, (_, r, p, pos) in enumerate(rows):
        hits[i] = (r, p, pos)
    vec = np.zeros(len(patterns) + 8, dtype=np.int64)
    vec[:len(patterns)] = c["pattern_hit_counts"]
    lay = sharding.counters_layout(len(patterns))
    vec[lay["hits"]] = c["hits"][0]
    vec[lay["records_hit"]] = c["records_hit"][0]
    vec[lay["records"]] = c["records"]
    vec[lay["bases"]] = c["bases"]
    return np.array([bool(f) for f in found]), hits, vec


def test_shard_bounds():
    assert sharding.shard_bounds(10, 4) == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert sharding.shard_bounds(0, 2) == [(0, 0), (0, 0)]
    assert sharding.shard_bounds(3, 8)[-1] == (3, 3)
    for n in (1, 7, 100, 12345):
        for w in (1, 2, 3, 8):
            b = sharding.shard_bounds(n, w)
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1


def test_merge_equals_unsharded_single_process():
    patterns, recs = _workload()
    f_all, h_all, v_all = _scan(patterns, recs)
    for world in (2, 3, 8):
        bounds = sharding.shard_bounds(len(recs), world)
        parts, vec = [], np.zeros_like(v_all)
        for lo, hi in bounds:
            f, h, v = _scan(patterns, recs[lo:hi])
            parts.append((f, h))
            vec += v
        f, h = sharding.merge_shards(parts, bounds)
        assert np.array_equal(f.astype(bool), f_all) and np.array_equal(h, h_all) and np.array_equal(vec, v_all)


def _worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        patterns, recs = _workload()
        bounds = sharding.shard_bounds(len(recs), world)
        lo, hi = bounds[rank]
        flags, hits, vec = _scan(patterns, recs[lo:hi])
        t = torch.from_numpy(vec.copy())
        sharding.all_reduce_counters(t)
        merged = sharding.gather_to_rank0(flags, hits, bounds)
        if rank == 0:
            f_all, h_all, v_all = _scan(patterns, recs)
            ok = (np.array_equal(merged[0].astype(bool), f_all) and np.array_equal(merged[1], h_all)
                  and np.array_equal(t.numpy(), v_all))
            open(os.path.join(out_dir, "result"), "w").write("ok" if ok else "mismatch")
        else:
            assert merged is None
            assert np.array_equal(t.numpy(), _scan(patterns, recs)[2])
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo(tmp_path):
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert open(tmp_path / "result").read() == "ok"

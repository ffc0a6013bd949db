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


class _FakeCommLib:
    """stands in for libmerkurio_hip.so's mk_comm_* entry points on one rank: `fail_at` names the call that
    fails on this rank ("bind", "init" or None)"""

    def __init__(self, rank, fail_at):
        self.rank, self.fail_at, self.calls = rank, fail_at, []

    def mk_last_error(self):
        return b"simulated failure"

    def mk_comm_unique_id(self, p):
        self.calls.append("unique_id")
        return -11 if self.fail_at == "bind" else 0

    def mk_comm_available(self):
        self.calls.append("available")
        return -11 if self.fail_at == "bind" else 0

    def mk_comm_init(self, handle, idp, rank, world):
        self.calls.append("init")
        return -11 if self.fail_at == "init" else 0


def _comm_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = []
        # (who fails, where): nobody; rank 1 cannot bind librccl; rank 0 cannot; rank 1's init comes back with an error
        for bad_rank, where in ((None, None), (1, "bind"), (0, "bind"), (1, "init")):
            lib = _FakeCommLib(rank, where if rank == bad_rank else None)
            ok, why = sharding.agree_on_communicator(lib, None, rank, world, "cpu")
            res.append((ok, why, lib.calls))
        open(os.path.join(out_dir, f"comm{rank}"), "w").write(repr(res))
    finally:
        dist.destroy_process_group()


def test_ranks_agree_on_the_rccl_communicator_or_give_up_together(tmp_path):
    """bench.py --gpus N: a rank that cannot bind librccl (or whose mk_comm_init fails) must land the whole job in
    the torch.distributed fallback -- never leave its peers waiting inside a collective init"""
    pytest.importorskip("torch")
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_comm_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (eval(open(tmp_path / f"comm{r}").read()) for r in (0, 1))
    assert [x[0] for x in r0] == [x[0] for x in r1] == [True, False, False, False]  # both ranks decide alike
    assert r0[0][2] == ["unique_id", "init"] and r1[0][2] == ["available", "init"]
    for case in (1, 2):  # a bind failure anywhere: NO rank enters the collective init
        assert "init" not in r0[case][2] and "init" not in r1[case][2]
        assert "simulated failure" in r0[case][1] and r0[case][1] == r1[case][1]
    assert "rank 1" in r0[3][1]

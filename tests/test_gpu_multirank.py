"""Multi-GPU path on the HIP kernels: records shard in contiguous ranges, no
data-path collective, one reduction of the counter vector at the end.

On the 1-GPU test box the ranks share device 0 (process group on gloo); what is under test is
the product path end to end: every rank scans ITS shard with mk_scan_device (hits mode, device
counters), results are merged in rank order and must equal the unsharded scan AND the oracle:
flags, ordered hits, counter vector.  mk_reduce_counters (RCCL, C ABI) is driven through ctypes
with two handles, and `python bench.py --gpus 2` must start its own ranks and exit 0."""
import json
import os
import random
import socket
import subprocess
import sys

import numpy as np
import pytest

import oracle_binding as ob

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _workload(seed=11, n_rec=6001, n_pat=300):
    rnd = random.Random(seed)
    raw = [bytes(rnd.choice(b"ACGT") for _ in range(31)) for _ in range(n_pat)]
    rc, patterns = ob.parse_pattern_list(raw, reverse_complement=True)
    assert rc == 0
    recs = []
    for i in range(n_rec):
        s = bytearray(rnd.choice(b"ACGT") for _ in range(rnd.choice([100, 150, 151, 250])))
        for _ in range(2 if i % 5 == 0 else 0):
            p = rnd.choice(patterns)
            k = rnd.randrange(0, len(s) - len(p) + 1)
            s[k:k + len(p)] = p
        recs.append(bytes(s))
    return patterns, recs


def _device_scan(mk, m, recs, device=0):
    """mk_scan_device on device-resident buffers -> (flags, ordered hits, counter vector)"""
    import torch
    lib = mk.load()
    dev = torch.device("cuda", device)
    data, off = mk.pack_records(recs)
    n_rec, n_bytes = len(recs), int(off[-1])
    d_seq = torch.zeros(n_bytes + 64, dtype=torch.uint8, device=dev)
    d_seq[:n_bytes] = torch.from_numpy(data[:n_bytes].copy()).to(dev)
    d_off = torch.from_numpy(off.astype(np.int64)).to(dev)
    d_flags = torch.zeros((n_rec + 7) // 4 * 4, dtype=torch.uint8, device=dev)
    cap = 1 << 16
    d_hits = torch.zeros(2 * cap, dtype=torch.int64, device=dev)
    d_nh = torch.zeros(1, dtype=torch.int64, device=dev)
    d_cnt = torch.zeros(len(m.patterns) + mk.MK_NUM_SUMMARY, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    rc = lib.mk_scan_device(m.handle, d_seq.data_ptr(), n_bytes, d_off.data_ptr(), n_rec, mk.MK_MODE_HITS,
                            d_flags.data_ptr(), d_hits.data_ptr(), cap, d_nh.data_ptr(), d_cnt.data_ptr(), st)
    assert rc == 0, lib.mk_last_error()
    torch.cuda.synchronize()
    n = int(d_nh.item())
    assert n <= cap
    hits = d_hits[:2 * n].cpu().numpy().view(mk.HIT_DTYPE).copy()
    assert lib.mk_order_hits(m.handle, hits.ctypes.data, n) == 0
    return d_flags[:n_rec].cpu().numpy() != 0, hits, d_cnt.cpu().numpy().copy(), d_cnt


def _oracle(patterns, recs):
    om = ob.Matcher(patterns, True, 0, False)
    keep, rows, c, found = ob.tag_records(om, recs, logging=True)
    hits = [(r, p, pos) for _, r, p, pos in rows]
    return [bool(f) for f in found], hits, c


def _worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from merkurio_amd import native as mk
    from merkurio_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = rank % torch.cuda.device_count()
        torch.cuda.set_device(dev)
        patterns, recs = _workload()
        m = mk.Matcher(patterns, device=dev)
        bounds = sharding.shard_bounds(len(recs), world)
        lo, hi = bounds[rank]
        flags, hits, vec, _ = _device_scan(mk, m, recs[lo:hi], dev)
        t = torch.from_numpy(vec.copy())
        sharding.all_reduce_counters(t)  # gloo here; RCCL (mk_comm_reduce_counters) with one GPU per rank
        merged = sharding.gather_to_rank0(flags, hits, bounds)
        if rank == 0:
            f_all, h_all, v_all, _ = _device_scan(mk, m, recs, dev)
            o_flags, o_hits, o_c = _oracle(patterns, recs)
            lay = sharding.counters_layout(len(patterns))
            got_hits = list(zip(merged[1]["rec"].tolist(), merged[1]["pat"].tolist(), merged[1]["pos"].tolist()))
            red = t.numpy()
            checks = {
                "flags==unsharded": bool(np.array_equal(merged[0].astype(bool), f_all)),
                "hits==unsharded": bool(np.array_equal(merged[1], h_all)),
                "counters==unsharded": bool(np.array_equal(np.delete(red, lay["candidates"]), np.delete(v_all, lay["candidates"]))),
                "flags==oracle": merged[0].astype(bool).tolist() == o_flags,
                "hits==oracle": got_hits == o_hits,
                "pattern_counts==oracle": red[lay["pattern_hit_counts"]].tolist() == o_c["pattern_hit_counts"],
                "scalars==oracle": (int(red[lay["hits"]]), int(red[lay["records_hit"]]), int(red[lay["records"]]), int(red[lay["bases"]]))
                                   == (o_c["hits"][0], o_c["records_hit"][0], o_c["records"], o_c["bases"]),
                "nonempty": len(got_hits) > 1000,
            }
            json.dump(checks, open(os.path.join(out_dir, "result.json"), "w"))
        else:
            assert merged is None
    finally:
        dist.destroy_process_group()


def test_two_ranks_shard_the_hip_scan(tmp_path):
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    checks = json.load(open(tmp_path / "result.json"))
    assert all(checks.values()), checks


def test_reduce_counters_c_abi():
    """mk_reduce_counters: two handles (both on device 0 here), each with the counter vector of its
    half of the batch; after the call BOTH device vectors and the host copy hold the whole-batch sum"""
    from merkurio_amd import native as mk
    from merkurio_amd import sharding
    patterns, recs = _workload(seed=12, n_rec=3000, n_pat=100)
    ms = [mk.Matcher(patterns, device=0), mk.Matcher(patterns, device=0)]
    bounds = sharding.shard_bounds(len(recs), 2)
    parts = [_device_scan(mk, m, recs[lo:hi]) for m, (lo, hi) in zip(ms, bounds)]
    _, _, v_all, _ = _device_scan(mk, ms[0], recs)
    n = len(patterns) + mk.MK_NUM_SUMMARY
    total = mk.reduce_counters(ms, [p[3].data_ptr() for p in parts], n)
    lay = sharding.counters_layout(len(patterns))
    keep = np.ones(n, dtype=bool)
    keep[lay["candidates"]] = False  # diagnostic: filter positives differ with the batch cut
    assert np.array_equal(total.astype(np.int64)[keep], v_all[keep])
    for p in parts:
        assert np.array_equal(p[3].cpu().numpy(), total.astype(np.int64))
    assert int(total[lay["hits"]]) > 500
    # argument errors come back as codes
    lib = mk.load()
    assert lib.mk_reduce_counters(None, 0, None, 0, None) == mk.MK_E_INVALID_ARG


def test_comm_single_rank_c_abi():
    """the one-process-per-GPU entry points with a world of one: id, init, in-place all-reduce"""
    import ctypes as C
    torch = pytest.importorskip("torch")
    from merkurio_amd import native as mk
    lib = mk.load()
    patterns, _ = _workload(seed=13, n_rec=1, n_pat=20)
    m = mk.Matcher(patterns, device=0)
    idb = np.zeros(mk.MK_COMM_ID_BYTES, dtype=np.uint8)
    assert lib.mk_comm_available() == 0, lib.mk_last_error()
    assert lib.mk_comm_unique_id(idb.ctypes.data) == 0, lib.mk_last_error()
    assert idb.any()
    assert lib.mk_comm_init(m.handle, idb.ctypes.data, 0, 1) == 0, lib.mk_last_error()
    t = torch.arange(48, dtype=torch.int64, device="cuda:0")
    assert lib.mk_comm_reduce_counters(m.handle, t.data_ptr(), 48, torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    assert t.cpu().tolist() == list(range(48))
    import ctypes
    seen = ctypes.c_int(0)  # what RCCL itself says (ncclCommCount): the figure bench.py prints as "RCCL saw N ranks"
    assert lib.mk_comm_size(m.handle, ctypes.byref(seen)) == 0 and seen.value == 1, lib.mk_last_error()
    assert lib.mk_comm_init(m.handle, idb.ctypes.data, 0, 1) == mk.MK_E_INVALID_ARG  # already initialised
    assert lib.mk_comm_destroy(m.handle) == 0


def _bench(*argv, timeout=900):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True,
                       timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_starts_its_own_ranks_and_strong_scaling_is_the_same_job():
    """`python bench.py --gpus 2` launches two ranks itself (rehearsal on one GPU) and the strong-
    scaled job is the same data set whatever N is: identical reduced counters at N=1 and N=2,
    single and paired (config-3 shape)."""
    common = ["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--scaling", "strong", "--total-records", "2000000",
              "--patterns", "2000"]
    one = _bench(*common)
    two = _bench("--gpus", "2", *common)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two.get("rehearsal") is True
    assert two["scaling"] == "strong" and two["config"]["records_per_gpu"] == 1000000
    keys = ("hits", "records_hit", "records", "bases")
    assert [one["summary"][k] for k in keys] == [two["summary"][k] for k in keys]
    assert one["summary"]["records"] == 2 * 2000000 and one["summary"]["hits"] > 0
    p1 = _bench(*common, "--paired")
    p2 = _bench("--gpus", "2", *common, "--paired")
    assert [p1["summary"][k] for k in keys] == [p2["summary"][k] for k in keys]
    assert p1["summary"]["records"] == 2 * 2 * 2000000
    weak = _bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--records", "1000000", "--patterns", "2000")
    assert weak["scaling"] == "weak" and weak["config"]["records_total"] == 2000000
    assert [weak["summary"][k] for k in keys] == [one["summary"][k] for k in keys]


def test_bench_default_job_at_two_ranks_reports_every_rank_and_the_multi_gpu_configs():
    """The driver's own command line at N = 2 (rehearsal: both ranks on the one GPU): the JSON line names every rank's
    device, kernel time and communicator size, and carries BASELINE's multi-GPU configurations -- config 3 (paired, pairs
    unsplit) and config 5 (500 k 21-mers) -- sharded over the same ranks."""
    out = _bench("--gpus", "2", "--steps", "2", "--warmup", "1")
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["records_total"] == 200_000_000
    assert [r["rank"] for r in out["ranks"]] == [0, 1] and all(r["records"] == 100_000_000 and r["kernel_ms_avg"] > 0 for r in out["ranks"])
    assert all("device" in r and "ncclCommCount" in r and "device_name" in r for r in out["ranks"])
    oc = out["other_configs"]
    assert isinstance(oc, list) and len(oc) == 2, oc
    c3, c5 = oc
    assert c3["workload"].startswith("config 3") and c3["launches_per_step"] == 2 and c3["n_gpus"] == 2 and len(c3["ranks"]) == 2
    assert 0 < c3["pairs_kept_in_the_job"] < 2 * 6_250_000 * 0.05
    assert c5["workload"].startswith("config 5") and c5["filter"]["in_lds"] == 0 and len(c5["ranks"]) == 2
    assert all(e["value_gbases_per_s"] > 0 and 0 < e["frac"] < 1 for e in oc)


def _agree_worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from merkurio_amd import native as mk
    from merkurio_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_dev = torch.cuda.device_count()
        dev = rank % n_dev
        torch.cuda.set_device(dev)
        lib = mk.load()
        m = mk.Matcher([b"ACGTACGTACGTACGTACGTA", b"TTTTTTTTTTTTTTTTTTTTT"], device=dev)
        ok, why = sharding.agree_on_communicator(lib, m.handle, rank, world, "cpu")
        total = None
        if ok:  # one GPU per rank: the in-place RCCL all-reduce of the counter vector through the C ABI
            n = len(m.patterns) + mk.MK_NUM_SUMMARY
            t = torch.full((n,), rank + 1, dtype=torch.int64, device=torch.device("cuda", dev))
            st = torch.cuda.current_stream().cuda_stream
            assert lib.mk_comm_reduce_counters(m.handle, t.data_ptr(), n, st) == 0, lib.mk_last_error()
            torch.cuda.synchronize()
            total = t.cpu().tolist()
            assert lib.mk_comm_destroy(m.handle) == 0
        json.dump({"ok": ok, "why": why, "n_dev": n_dev, "total": total}, open(os.path.join(out_dir, f"agree{rank}.json"), "w"))
    finally:
        dist.destroy_process_group()


def test_ranks_set_up_the_rccl_communicator_or_fall_back_together(tmp_path):
    """bench.py's protocol (sharding.agree_on_communicator) with the REAL library, two ranks: with one GPU per rank the
    C ABI's RCCL communicator comes up and mk_comm_reduce_counters sums the vectors; on a 1-GPU box RCCL refuses two
    ranks on one device -- then BOTH ranks must come back with the same refusal (the job falls back to
    torch.distributed), none may be left waiting in the collective init"""
    pytest.importorskip("torch")
    import time

    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.spawn(_agree_worker, args=(2, port, str(tmp_path)), nprocs=2, join=False)
    deadline = time.time() + 240
    while not ctx.join(timeout=5):
        if time.time() > deadline:
            for p in ctx.processes:
                p.kill()
            pytest.fail("a rank is still waiting inside the communicator set-up after 240 s")
    r0, r1 = (json.load(open(tmp_path / f"agree{r}.json")) for r in (0, 1))
    assert r0["ok"] == r1["ok"] and r0["why"] == r1["why"], (r0, r1)
    if r0["n_dev"] >= 2:
        assert r0["ok"], r0
        assert r0["total"] == r1["total"] and set(r0["total"]) == {3}
    else:
        assert not r0["ok"] and r0["why"], r0


_FAKE_CHILD = r'''
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
os.environ["MERKURIO_SYSTEM_HIP"] = "1"  # no torch in this process: nothing has mapped the real librccl
from merkurio_amd import native as mk
lib = mk.load()
fake = C.CDLL("librccl.so.1")
assert hasattr(fake, "fake_rccl_live_comms"), "the real librccl was bound"
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
ms = [mk.Matcher([b"ACGTACGTACGTACGTACGTA"], device=0) for _ in range(3)]
n = 9
vec = []
for i in range(3):
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), n * 8) == 0
    v = np.arange(n, dtype=np.uint64) * (i + 1)
    assert hip.hipMemcpy(p, v.ctypes.data, n * 8, 1) == 0
    vec.append(p)
handles = (C.c_void_p * 3)(*[m.handle for m in ms])
ptrs = (C.c_void_p * 3)(*vec)
out = np.zeros(n, dtype=np.uint64)
def reduce():
    return lib.mk_reduce_counters(handles, 3, ptrs, n, out.ctypes.data)
# 1. ncclCommInitAll fails (having written garbage into the first slot): MK_E_RCCL, nothing cached, nothing destroyed
os.environ["FAKE_RCCL_FAIL"] = "initall"
assert reduce() == mk.MK_E_RCCL and b"CommInitAll" in lib.mk_last_error(), lib.mk_last_error()
assert fake.fake_rccl_live_comms() == 0 and fake.fake_rccl_calls(2) == 0
# 2. the grouped all-reduce fails: MK_E_RCCL, the communicator set is destroyed, not kept
os.environ["FAKE_RCCL_FAIL"] = "allreduce"
assert reduce() == mk.MK_E_RCCL and b"AllReduce" in lib.mk_last_error()
assert fake.fake_rccl_live_comms() == 0 and fake.fake_rccl_calls(0) == 2
os.environ["FAKE_RCCL_FAIL"] = "groupend"
assert reduce() == mk.MK_E_RCCL and b"GroupEnd" in lib.mk_last_error()
assert fake.fake_rccl_live_comms() == 0 and fake.fake_rccl_calls(0) == 3
# 3. RCCL works again: a fresh set is created, kept and reused
os.environ["FAKE_RCCL_FAIL"] = ""
for v, p in zip(range(3), vec):  # (the failed attempts had already added the device-local vectors together)
    a = np.arange(n, dtype=np.uint64) * (v + 1)
    assert hip.hipMemcpy(p, a.ctypes.data, n * 8, 1) == 0
assert reduce() == 0, lib.mk_last_error()
assert out.tolist() == (np.arange(n) * 6).tolist()
assert fake.fake_rccl_live_comms() == 1 and fake.fake_rccl_calls(0) == 4
assert reduce() == 0 and fake.fake_rccl_calls(0) == 4  # cached
# 4. one process per GPU: a failing ncclCommInitRank leaves the handle without a communicator; a good one reports its size
ident = (C.c_uint8 * 128)()
assert lib.mk_comm_unique_id(ident) == 0
os.environ["FAKE_RCCL_FAIL"] = "initrank"
assert lib.mk_comm_init(ms[0].handle, ident, 0, 1) == mk.MK_E_RCCL
assert lib.mk_comm_reduce_counters(ms[0].handle, vec[0], n, None) == mk.MK_E_INVALID_ARG
os.environ["FAKE_RCCL_FAIL"] = ""
assert lib.mk_comm_init(ms[0].handle, ident, 0, 1) == 0
k = C.c_int()
assert lib.mk_comm_size(ms[0].handle, C.byref(k)) == 0 and k.value == 1
os.environ["FAKE_RCCL_FAIL"] = "allreduce"
assert lib.mk_comm_reduce_counters(ms[0].handle, vec[0], n, None) == mk.MK_E_RCCL
os.environ["FAKE_RCCL_FAIL"] = ""
assert lib.mk_comm_reduce_counters(ms[0].handle, vec[0], n, None) == 0
assert lib.mk_comm_destroy(ms[0].handle) == 0 and fake.fake_rccl_live_comms() == 1
print("fake-rccl ok")
'''


def test_reduction_failure_modes_are_total(tmp_path):
    """mk_reduce_counters / mk_comm_* against a fake librccl with failure injection (tests/fake_rccl/fake_rccl.c, bound
    through reduce.cpp's dlopen-by-soname seam in a child process): a failing ncclCommInitAll caches nothing, a failing
    grouped ncclAllReduce / ncclGroupEnd destroys the communicator set instead of keeping it, the next call starts
    over and succeeds; a failing ncclCommInitRank leaves the handle without a communicator.  (The multi-device
    branches themselves have never run on hardware: the pool's boxes have one GPU)"""
    so = tmp_path / "librccl.so.1"
    subprocess.run(["gcc", "-shared", "-fPIC", "-O1", "-Wl,-soname,librccl.so.1", "-o", str(so),
                    os.path.join(ROOT, "tests", "fake_rccl", "fake_rccl.c")], check=True)
    env = dict(os.environ, LD_LIBRARY_PATH=f"{tmp_path}:{os.environ.get('LD_LIBRARY_PATH', '')}")
    r = subprocess.run([sys.executable, "-c", _FAKE_CHILD, ROOT], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "fake-rccl ok" in r.stdout, r.stdout + r.stderr

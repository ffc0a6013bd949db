#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native k-mer scanner.

One "step" = one pass of the hot path (mk_scan_device, any-hit flags = `merkurio extract`
without logging) over one device-resident batch of synthetic reads.  Workload at N=1: the
configuration BASELINE.json's metric is quoted on: 100 M x 150 bp reads, 10 k 31-mers
(Aho-Corasick by the reference's selection rule).  N>1: one process per GPU, every rank scans
its own 100 M-read shard (weak scaling, no data-path collective); the only collective is the
RCCL all-reduce of the hit-count / summary vector at the end of the job.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--records R] [--read-len L]
                    [--patterns P] [--k K] [--no-cpu-baseline]

Prints ONE JSON line on rank 0 (contract in the task statement), with `roofline` (HBM) and
`cpu_baseline` objects.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)


def measured_traffic(kernel, n_rec, read_len, n_pat):
    """HBM bytes per launch of the dominant kernel from the newest committed PMC profile of this
    exact workload (profiles/traffic_rNN.json: rocprofv3 FETCH_SIZE/WRITE_SIZE passes, gfx950
    correction applied); None if no such profile is committed."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "traffic_r*.json"))):
        try:
            j = json.load(open(f))
        except (OSError, ValueError):
            continue
        if (j.get("kernel"), j.get("records_per_gpu"), j.get("read_len"), j.get("patterns")) == \
                (kernel, n_rec, read_len, n_pat):
            best = (j["hbm_bytes_per_launch"], os.path.relpath(f, ROOT))
    return best


def make_patterns(n, k, seed=0x4D65724B):
    rng = np.random.default_rng(seed)
    codes = rng.integers(0, 4, size=(int(n * 1.01) + 8, k), dtype=np.uint8)
    arr = np.frombuffer(b"ACGT", dtype=np.uint8)[codes]
    return [arr[i].tobytes() for i in range(arr.shape[0])]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--records", type=int, default=100_000_000, help="reads per GPU")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--patterns", type=int, default=10_000)
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--plant-every", type=int, default=100, help="1 in N reads carries a planted k-mer")
    ap.add_argument("--rc", action="store_true", help="add reverse complements to the pattern list (-r)")
    ap.add_argument("--mode", choices=["any", "hits"], default="any",
                    help="any = per-record flags (extract without logging, the headline); hits = also emit every "
                         "(record, pattern, position) tuple (extract/tag with logging)")
    ap.add_argument("--no-counters", action="store_true", help="diagnostic: scan without the device counter vector")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the baseline sample")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from merkurio_amd import native as mk
    from merkurio_amd import sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    # rehearsal hook: MERKURIO_BENCH_BACKEND=gloo runs several ranks on fewer GPUs (rank -> device
    # round-robin); the measured configuration is one rank per GPU over nccl (= RCCL)
    backend = os.environ.get("MERKURIO_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    use_dist = "RANK" in os.environ and "MASTER_PORT" in os.environ  # launched by torch.distributed.run
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    # ---- pattern set (identical on every rank) and matcher
    raw = make_patterns(args.patterns, args.k)
    patterns = mk.parse_pattern_list(kmer_seq=raw)[:args.patterns]
    assert len(patterns) == args.patterns
    if args.rc:
        patterns = mk.parse_pattern_list(kmer_seq=patterns, reverse_complement=True)
    m = mk.Matcher(patterns, device=dev_index)
    assert m.use_ac == mk.recommend_aho_corasick(patterns)
    lib = mk.load()

    # ---- device-resident synthetic batch (records shard = rank)
    n_rec, L = args.records, args.read_len
    n_bytes = n_rec * L
    seed = 0x4D65724B7572696F + rank
    d_seq = torch.empty(n_bytes + 64, dtype=torch.uint8, device=dev)
    d_off = torch.empty(n_rec + 1, dtype=torch.int64, device=dev)
    d_flags = torch.empty((n_rec + 7) // 4 * 4, dtype=torch.uint8, device=dev)
    d_nh = torch.zeros(1, dtype=torch.int64, device=dev)
    d_cnt = torch.zeros(len(patterns) + mk.MK_NUM_SUMMARY, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream()
    st = stream.cuda_stream
    rc = lib.mk_synth_reads_device(m.handle, seed, n_rec, L, args.plant_every, d_seq.data_ptr(), d_off.data_ptr(), st)
    assert rc == 0, lib.mk_last_error()
    torch.cuda.synchronize()

    emit = args.mode == "hits"
    hits_cap = max(1 << 20, 4 * n_rec // max(1, args.plant_every)) if emit else 0
    d_hits = torch.empty(2 * hits_cap, dtype=torch.int64, device=dev) if emit else None

    def step():
        rc = lib.mk_scan_device(m.handle, d_seq.data_ptr(), n_bytes, d_off.data_ptr(), n_rec,
                                mk.MK_MODE_HITS if emit else mk.MK_MODE_ANY, d_flags.data_ptr(),
                                d_hits.data_ptr() if emit else None, hits_cap, d_nh.data_ptr(),
                                None if args.no_counters else d_cnt.data_ptr(), st)
        if rc != 0:
            raise RuntimeError(lib.mk_last_error().decode())

    def barrier():
        if use_dist:
            if backend == "nccl":
                dist.barrier(device_ids=[dev_index])
            else:
                dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    d_cnt.zero_()
    m.enable_timing(args.steps)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sharding.all_reduce_counters(d_cnt)  # the job's only collective: hit-count / summary vector (RCCL over xGMI)
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    kernel_ms = m.kernel_times_ms()
    cnt = d_cnt.cpu().numpy()
    summ = cnt[len(patterns):]

    if rank == 0:
        total_bases = world * n_bytes * args.steps
        value = total_bases / dt / 1e9
        k_avg_ms = float(np.mean(kernel_ms))
        algo_bytes = n_bytes * 1 + n_rec * 9  # 1 B/base + u64 offset + 1 B flag per record
        achieved = algo_bytes / (k_avg_ms * 1e-3) / 1e9
        info = dict(m.filter_info(), **m.filter_mode())
        out = {
            "metric": "Gbases/s scanned (150bp FASTQ, 10k 31-mers); % HBM roofline at 1/2/4/8 GPUs",
            "value": round(value, 3),
            "unit": "Gbases/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": f"extract ({'all hit tuples' if emit else 'any-hit flags'}): {n_rec} x {L} bp synthetic reads per GPU, "
                            f"{len(patterns)} {args.k}-mers, Aho-Corasick semantics, 1/{args.plant_every} reads planted",
                "records_per_gpu": n_rec, "read_len": L, "patterns": len(patterns), "k": args.k,
                "sharding": f"records x{world}", "kernel": m.kernel_name, "filter": info,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": None,
                "traffic_source": None,
                "kernel_ms_avg": round(k_avg_ms, 4),
                "kernel_ms_median": round(float(np.median(kernel_ms)), 4),
                "kernel_ms_min": round(float(np.min(kernel_ms)), 4),
                "algorithmic_bytes_per_launch": algo_bytes,
            },
            "summary": {"hits": int(summ[mk.MK_SUM_HITS]), "records_hit": int(summ[mk.MK_SUM_RECORDS_HIT]),
                        "records": int(summ[mk.MK_SUM_RECORDS]), "bases": int(summ[mk.MK_SUM_BASES]),
                        "filter_candidates": int(summ[mk.MK_SUM_CANDIDATES])},
        }
        tr = measured_traffic(m.kernel_name, n_rec, L, len(patterns))
        if tr:
            out["roofline"]["traffic"], out["roofline"]["traffic_source"] = tr
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(mk, m, patterns, seed, n_rec, L, args.plant_every, d_flags,
                                               args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


def host_cores():
    """cores this process may really use: CPU affinity, capped by the cgroup CPU quota"""
    n = len(os.sched_getaffinity(0))
    for quota_file, period_file in (("/sys/fs/cgroup/cpu.max", None),
                                    ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us")):
        try:
            if period_file is None:
                quota, period = open(quota_file).read().split()[:2]
            else:
                quota, period = open(quota_file).read().strip(), open(period_file).read().strip()
            if quota not in ("max", "-1"):
                n = min(n, max(1, int(quota) // int(period)))
        except (OSError, ValueError):
            pass
    return min(n, int(os.environ.get("MERKURIO_BENCH_CPU_THREADS", "16")))  # a 1-GPU box's CPU share is 16


def cpu_baseline(mk, m, patterns, seed, n_rec, L, plant_every, d_flags, target_s):
    """The reference's CPU path for this workload, restated (oracle, kind "port"): dense
    Aho-Corasick DFA, first-hit break per record (src/cmd_extract.rs:332-335), 1 thread --
    the reference matcher is single-threaded.  Timed on a bounded sample (the first M reads
    of rank 0's shard); the GPU flags for the same reads are checked against it."""
    import oracle_binding as ob
    lib = mk.load()
    om = ob.Matcher(patterns, True, 0, False)

    def run(M):
        seq = np.zeros(M * L, dtype=np.uint8)
        off = np.zeros(M + 1, dtype=np.uint64)
        assert lib.mk_synth_reads_host(m.handle, seed, 0, M, L, plant_every, seq.ctypes.data, off.ctypes.data) == 0
        t0 = time.perf_counter()
        keep, _ = ob.extract_single_packed(om, seq, off, logging=False, invert=False)
        return time.perf_counter() - t0, keep

    probe = min(n_rec, 200_000)
    t, keep = run(probe)
    M = int(min(n_rec, max(probe, probe * target_s / max(t, 1e-6))))
    if M > probe:
        t, keep = run(M)
    gpu = d_flags[:M].cpu().numpy()
    res = {"value": round(M * L / t / 1e9, 4), "unit": "Gbases/s", "cores": 1, "kind": "port",
           "sample": f"first {M} reads ({M * L / 1e6:.0f} Mbases) of the same synthetic workload, "
                     f"oracle Aho-Corasick DFA with first-hit break, {t:.1f} s",
           "gpu_flags_match_cpu": bool(np.array_equal(gpu != 0, keep != 0)), "records_kept": int(keep.sum())}
    # context only (SURVEY.md §8d ii): the same port on every host core this process may use,
    # contiguous record shards, one thread each (ctypes releases the GIL; the DFA is read-only)
    cores = host_cores()
    if cores > 1:
        from concurrent.futures import ThreadPoolExecutor
        per = int(min(n_rec // cores, max(1, M // 4), 2_000_000))

        def make(c):
            seq = np.zeros(per * L, dtype=np.uint8)
            off = np.zeros(per + 1, dtype=np.uint64)
            assert lib.mk_synth_reads_host(m.handle, seed, c * per, per, L, plant_every, seq.ctypes.data, off.ctypes.data) == 0
            return seq, off

        with ThreadPoolExecutor(cores) as ex:
            bufs = list(ex.map(make, range(cores)))
            t0 = time.perf_counter()
            kept = list(ex.map(lambda b: int(ob.extract_single_packed(om, b[0], b[1], logging=False, invert=False)[0].sum()), bufs))
            ta = time.perf_counter() - t0
        res["all_cores"] = {"value": round(cores * per * L / ta / 1e9, 4), "unit": "Gbases/s", "cores": cores,
                            "sample": f"first {cores * per} reads in {cores} contiguous shards, {ta:.1f} s",
                            "records_kept": int(sum(kept))}
    return res


if __name__ == "__main__":
    main()

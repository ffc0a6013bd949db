#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native k-mer scanner.

One "step" = one pass of the hot path (mk_scan_device, any-hit flags = `merkurio extract`
without logging) over one device-resident batch of synthetic reads.  Workload at N=1: the
configuration BASELINE.json's metric is quoted on: 100 M x 150 bp reads, 10 k 31-mers
(Aho-Corasick by the reference's selection rule).

N>1: one process per GPU, records sharded in contiguous ranges (pairs never split), no
data-path collective; the only collective is the RCCL all-reduce of the hit-count / summary
vector at the end of the job (mk_comm_reduce_counters of the C ABI).

    python bench.py [--gpus N] [--steps K] [--warmup W]
                    [--scaling weak|strong] [--total-records R] [--paired]
                    [--records R] [--read-len L] [--patterns P] [--k K] [--mode any|hits] ...

  --scaling weak    (default) every rank scans --records reads: the job grows with N
  --scaling strong  the job is --total-records reads (or pairs with --paired) divided over the
                    ranks: BASELINE config 3 = --scaling strong --paired --total-records 50000000,
                    config 5 = --scaling strong --total-records 100000000 --read-len 250
                    --patterns 500000 --k 21

`python bench.py --gpus N` starts its own N ranks (torch.distributed.run as a child process,
before this process has touched a GPU); under an external launcher (WORLD_SIZE == N already)
it runs as one rank.  With fewer GPUs than ranks (a 1-GPU box) it runs in REHEARSAL mode:
ranks share devices round-robin, process group on gloo, counter reduction through
torch.distributed -- functional check only, marked "rehearsal": true in the output.

Prints ONE JSON line on rank 0 (contract in the task statement), with `roofline` (HBM) and,
at N=1, `cpu_baseline` objects.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)
KERNEL_SOURCES = ["merkurio_amd/csrc/scan_kernel_impl.hpp", "merkurio_amd/csrc/filter.hpp"]


def kernel_source_hash():
    """sha256 over the files that define the scan kernel: a PMC traffic figure is only valid for
    the kernel body it was measured on"""
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        with open(os.path.join(ROOT, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def measured_traffic(kernel, n_rec, read_len, n_pat, plant_every=100):
    """HBM bytes per launch of the dominant kernel from the newest committed PMC profile of this
    exact workload AND this exact kernel source (profiles/traffic_rNN.json: rocprofv3
    FETCH_SIZE/WRITE_SIZE passes, gfx950 correction applied).  -> (bytes | None, source note)"""
    import glob
    cur = kernel_source_hash()
    best, stale = None, None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "traffic_r*.json"))):
        try:
            j = json.load(open(f))
        except (OSError, ValueError):
            continue
        if (j.get("kernel"), j.get("records_per_gpu"), j.get("read_len"), j.get("patterns"), j.get("plant_every", 100)) != \
                (kernel, n_rec, read_len, n_pat, plant_every):
            continue
        if j.get("kernel_source_sha16") == cur:
            best = (j["hbm_bytes_per_launch"], os.path.relpath(f, ROOT))
        else:
            stale = os.path.relpath(f, ROOT)
    if best:
        return best
    if stale:
        return None, f"refused: {stale} was measured on a different kernel source (now {cur})"
    return None, None


def make_patterns(n, k, seed=0x4D65724B):  # (the pattern set of a run never depends on anything but n, k, seed)
    import numpy as np
    rng = np.random.default_rng(seed)
    codes = rng.integers(0, 4, size=(int(n * 1.01) + 8, k), dtype=np.uint8)
    arr = np.frombuffer(b"ACGT", dtype=np.uint8)[codes]
    return [arr[i].tobytes() for i in range(arr.shape[0])]


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--records", type=int, default=100_000_000, help="weak scaling: reads (or pairs) per GPU")
    ap.add_argument("--total-records", type=int, default=0, help="strong scaling: reads (or pairs) of the whole job")
    ap.add_argument("--paired", action="store_true", help="paired-end: two mate batches per step, pairs never split")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--patterns", type=int, default=10_000)
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--plant-every", type=int, default=100, help="1 in N reads carries a planted k-mer")
    ap.add_argument("--rc", action="store_true", help="add reverse complements to the pattern list (-r)")
    ap.add_argument("--random-kmers", action="store_true",
                    help="all k-mers uniform random (default: half of them sampled from reads of the job, SURVEY.md 8d)")
    ap.add_argument("--mode", choices=["any", "hits"], default="any",
                    help="any = per-record flags (extract without logging, the headline); hits = also emit every "
                         "(record, pattern, position) tuple (extract/tag with logging)")
    ap.add_argument("--ragged", type=int, default=0, metavar="SPREAD",
                    help="diagnostic: record lengths uniform in [read_len - SPREAD, read_len + SPREAD] over the same bytes "
                         "(trimmed reads: the record lookup of a verified occurrence can no longer guess its index)")
    ap.add_argument("--density-hint", type=int, default=-1,
                    help="diagnostic: records hit per 1000 told to the library instead of what the warm-up saw")
    ap.add_argument("--with-offsets", action="store_true",
                    help="diagnostic: do not tell the library that all reads have one length (the record of an occurrence is looked up in the offsets)")
    ap.add_argument("--no-rec-index", action="store_true", help="diagnostic with --ragged: do not tell the library that lengths vary")
    ap.add_argument("--no-counters", action="store_true", help="diagnostic: scan without the device counter vector")
    ap.add_argument("--no-order", action="store_true", help="diagnostic with --mode hits: leave the tuples unordered")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="N = 1, default workload only: skip the short runs of the other BASELINE configurations that are "
                         "appended to the JSON line as other_configs")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the baseline sample")
    # filter-geometry tuning hooks (mk_matcher_options; results never depend on them)
    ap.add_argument("--force-stride", type=int, default=0)
    ap.add_argument("--force-global-filter", action="store_true")
    ap.add_argument("--gbloom-log2-blocks", type=int, default=0)
    ap.add_argument("--tile-run", type=int, default=0)
    ap.add_argument("--gbloom-kib", type=int, default=0)
    return ap.parse_args(argv)


def spawn_ranks(args):
    """--gpus N without an external launcher: start N ranks as a child job and relay its output.
    Nothing in this (parent) process has touched HIP or torch.cuda."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    env["MASTER_ADDR"] = "127.0.0.1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse_args()
    if args.ragged and (args.paired or args.ragged >= args.read_len):
        raise SystemExit("--ragged SPREAD: single-end only, SPREAD < --read-len")
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world_env != args.gpus:
        if "RANK" in os.environ:
            raise SystemExit(f"--gpus {args.gpus} under a launcher with WORLD_SIZE={world_env}")
        raise SystemExit(spawn_ranks(args))

    import numpy as np
    import torch
    import torch.distributed as dist
    from merkurio_amd import native as mk
    from merkurio_amd import sharding

    world = world_env if args.gpus > 1 else 1
    rank = int(os.environ.get("RANK", "0")) if world > 1 else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if world > 1 else 0
    n_dev = torch.cuda.device_count()
    if n_dev < 1:
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (the scan path has no CPU fallback)")
    rehearsal = world > n_dev  # fewer GPUs than ranks: functional rehearsal, ranks share devices
    dev_index = local_rank % n_dev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    use_dist = world > 1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)  # backend "nccl" IS RCCL on ROCm

    # ---- pattern set (identical on every rank) and matcher.  SURVEY.md 8d / the reference's own recipe
    # (benchmarks/scripts/02-generate-kmers.sh:24-39): half of the k-mers are substrings of distinct reads of the job
    # at uniform offsets, half uniform random (--random-kmers: all random, the round-1/2 recipe)
    L = args.read_len
    seed = 0x4D65724B7572696F
    lib = mk.load()
    raw = make_patterns(args.patterns, args.k)
    if not args.random_kmers and args.k <= L:
        job_reads = (args.total_records or args.records) if args.scaling == "strong" else args.records * world
        rng = np.random.default_rng(0x6B6D6572)
        n_from = args.patterns // 2
        picks = np.unique(rng.integers(0, job_reads, size=n_from + n_from // 8 + 8))[:n_from]
        offs = rng.integers(0, L - args.k + 1, size=len(picks))
        tmp = mk.Matcher([b"A" * args.k], device=dev_index)  # (only its handle: the generator plants nothing here)
        buf, one_off = np.zeros(L, dtype=np.uint8), np.zeros(2, dtype=np.uint64)
        sampled = []
        for r, o in zip(picks.tolist(), offs.tolist()):
            mk._check(lib.mk_synth_reads_host(tmp.handle, seed, r, 1, L, 0, buf.ctypes.data, one_off.ctypes.data))
            sampled.append(buf[o:o + args.k].tobytes())
        tmp.close()
        raw = sampled + raw[len(sampled):]
    patterns = mk.parse_pattern_list(kmer_seq=raw)[:args.patterns]
    assert len(patterns) == args.patterns
    if args.rc:
        patterns = mk.parse_pattern_list(kmer_seq=patterns, reverse_complement=True)
    options = None
    if args.force_stride or args.force_global_filter or args.gbloom_log2_blocks or args.tile_run or args.gbloom_kib:
        options = dict(force_stride=args.force_stride, force_global_filter=args.force_global_filter,
                       gbloom_log2_blocks=args.gbloom_log2_blocks, tile_run=args.tile_run, gbloom_kib=args.gbloom_kib)
    m = mk.Matcher(patterns, device=dev_index, options=options)
    assert m.use_ac == mk.recommend_aho_corasick(patterns)
    if args.ragged and not args.no_rec_index:
        mk._check(lib.mk_matcher_hint_record_lengths(m.handle, 0))
    if not args.ragged and not args.with_offsets:
        # the synthetic reads all have --read-len bases: fixed-length batches (what mk_scan_batch finds out by
        # itself from a FASTQ of untrimmed reads); the offsets array is still built and still counted in the
        # algorithmic bytes, the kernel just has no use for it
        mk._check(lib.mk_matcher_set_fixed_record_length(m.handle, args.read_len))

    # ---- this rank's shard of the job: contiguous range of records (pairs), in units of 16 so
    # that every shard starts on a block of the counter-based generator
    if args.scaling == "strong":
        total = (args.total_records or args.records) // 16 * 16
        lo, hi = [16 * b for b in sharding.shard_bounds(total // 16, world)[rank]]
    else:
        total = args.records * world
        lo, hi = rank * args.records, (rank + 1) * args.records
        if (lo * L) % 32:
            raise SystemExit("weak scaling with N > 1 needs --records * --read-len divisible by 32")
    n_rec = hi - lo
    n_bytes = n_rec * L
    n_mates = 2 if args.paired else 1
    stream = torch.cuda.current_stream()
    st = stream.cuda_stream
    mates = []
    for f in range(n_mates):
        d_seq = torch.empty(n_bytes + 64, dtype=torch.uint8, device=dev)
        d_off = torch.empty(n_rec + 1, dtype=torch.int64, device=dev)
        d_flags = torch.empty((n_rec + 7) // 4 * 4, dtype=torch.uint8, device=dev)
        rc = lib.mk_synth_reads_device_range(m.handle, seed + f, lo, n_rec, L, args.plant_every, d_seq.data_ptr(),
                                             d_off.data_ptr(), st)
        assert rc == 0, lib.mk_last_error()
        if args.ragged:  # same bytes, cut into records of random length; the batch ends with the last whole record
            g = torch.Generator(device="cpu").manual_seed(1234 + f)
            lens = torch.randint(L - args.ragged, L + args.ragged + 1, (n_rec,), generator=g, dtype=torch.int64)
            off = torch.zeros(n_rec + 1, dtype=torch.int64)
            torch.cumsum(lens, 0, out=off[1:])
            keep = int(torch.searchsorted(off, torch.tensor([n_bytes]), right=True).item()) - 1
            ragged_shape = (keep, int(off[keep].item()))
            d_off = off[:keep + 1].to(dev)
        mates.append((d_seq, d_off, d_flags))
    if args.ragged:
        n_rec, n_bytes = ragged_shape
        total = n_rec * world
    d_nh = torch.zeros(1, dtype=torch.int64, device=dev)
    d_cnt = torch.zeros(len(patterns) + mk.MK_NUM_SUMMARY, dtype=torch.int64, device=dev)
    d_keep = torch.empty_like(mates[0][2]) if args.paired else None
    torch.cuda.synchronize()

    emit = args.mode == "hits"
    hits_cap = max(1 << 20, 4 * n_rec // max(1, args.plant_every)) if emit else 0
    d_hits = torch.empty(2 * hits_cap, dtype=torch.int64, device=dev) if emit else None

    order_s = []  # hits mode: wall time of every emission-order call (it waits for the stream once by itself)

    def step():
        for d_seq, d_off, d_flags in mates:
            rc = lib.mk_scan_device(m.handle, d_seq.data_ptr(), n_bytes, d_off.data_ptr(), n_rec,
                                    mk.MK_MODE_HITS if emit else mk.MK_MODE_ANY, d_flags.data_ptr(),
                                    d_hits.data_ptr() if emit else None, hits_cap, d_nh.data_ptr(),
                                    None if args.no_counters else d_cnt.data_ptr(), st)
            if rc != 0:
                raise RuntimeError(lib.mk_last_error().decode())
            if emit and not args.no_order:
                # the tuples in the reference's emission order (src/cmd_extract.rs:338-351): part of the step --
                # the host reads the tuple count, then the device bins and sorts the tuples in place
                nh = int(d_nh.item())
                if nh > hits_cap:
                    raise RuntimeError(f"{nh} tuples do not fit the buffer of {hits_cap}")
                t_o = time.perf_counter()
                mk._check(lib.mk_order_hits_device(m.handle, d_hits.data_ptr(), nh, st))
                torch.cuda.synchronize()
                order_s.append(time.perf_counter() - t_o)
        if args.paired:  # a pair is kept if either mate hits (src/cmd_extract.rs:600-606)
            torch.bitwise_or(mates[0][2], mates[1][2], out=d_keep)

    def barrier():
        if use_dist:
            if rehearsal:
                dist.barrier()
            else:
                dist.barrier(device_ids=[dev_index])
        torch.cuda.synchronize()

    # ---- the job's only collective, through the C ABI (RCCL); torch.distributed carries the id
    reduce_via = "none (1 GPU)"
    if use_dist and not rehearsal:
        ok, why = sharding.agree_on_communicator(lib, m.handle, rank, world, dev)
        if ok:
            import ctypes
            seen = ctypes.c_int(0)  # what RCCL itself says about the communicator (ncclCommCount)
            mk._check(lib.mk_comm_size(m.handle, ctypes.byref(seen)))
            reduce_via = f"mk_comm_reduce_counters (RCCL ncclAllReduce, C ABI; RCCL saw {seen.value} ranks)"
        else:  # same sum either way; say which path ran and why
            reduce_via = f"torch.distributed all_reduce over RCCL (C-ABI communicator unavailable: {why})"
    elif use_dist:
        reduce_via = "torch.distributed all_reduce over gloo (rehearsal)"

    def reduce_counters():
        if not use_dist:
            return
        if reduce_via.startswith("mk_comm"):
            mk._check(lib.mk_comm_reduce_counters(m.handle, d_cnt.data_ptr(), d_cnt.numel(), st))
        elif rehearsal:
            t = d_cnt.cpu()
            sharding.all_reduce_counters(t)
            d_cnt.copy_(t)
        else:
            sharding.all_reduce_counters(d_cnt)

    for _ in range(args.warmup):
        step()
    barrier()
    if args.warmup and not args.no_counters:
        # feed the hit density the warm-up steps saw back to the library (it picks cacheable stream
        # loads for hit-dense text); a host that scans batch after batch does the same with mk_scan_batch
        w = d_cnt.cpu().numpy()[len(patterns):]
        if w[mk.MK_SUM_RECORDS]:
            seen = int(w[mk.MK_SUM_RECORDS_HIT]) * 1000 // int(w[mk.MK_SUM_RECORDS])
            mk._check(lib.mk_matcher_hint_hit_density(m.handle, seen if args.density_hint < 0 else args.density_hint))
    # warm-up of the collective as well: the first all-reduce on a new communicator sets up its
    # channels (tens of ms), which belongs to the warm-up like the first launches do; the vector is
    # cleared afterwards, the timed job reduces its own counters once
    reduce_counters()
    torch.cuda.synchronize()
    d_cnt.zero_()
    m.enable_timing(args.steps * n_mates)
    del order_s[:]
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    reduce_counters()  # hit-count / summary vector (RCCL over xGMI), once per job
    barrier()
    dt = time.perf_counter() - t0
    dt_local = dt
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    kernel_ms = m.kernel_times_ms()
    cnt = d_cnt.cpu().numpy()
    summ = cnt[len(patterns):]
    # what every rank saw of its own shard (N > 1: gathered on rank 0 -- the first real multi-GPU run should say which rank
    # sat on which device, what its kernel took and how many ranks RCCL counted in its communicator)
    comm_ranks = None
    if reduce_via.startswith("mk_comm"):
        import ctypes
        seen_n = ctypes.c_int(0)
        mk._check(lib.mk_comm_size(m.handle, ctypes.byref(seen_n)))
        comm_ranks = seen_n.value
    mine = {"rank": rank, "local_rank": local_rank, "device": dev_index, "device_name": torch.cuda.get_device_name(dev_index),
            "records": n_rec, "kernel_ms_avg": round(float(np.mean(kernel_ms)), 4), "kernel_ms_min": round(float(np.min(kernel_ms)), 4),
            "kernel_ms_max": round(float(np.max(kernel_ms)), 4), "step_wall_ms_this_rank": round(dt_local / args.steps * 1e3, 4),
            "ncclCommCount": comm_ranks}
    per_rank = [mine]
    if use_dist:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    if rank == 0:
        total_bases = (n_bytes * world if args.ragged else total * L) * n_mates * args.steps
        value = total_bases / dt / 1e9
        k_avg_ms = float(np.mean(kernel_ms))
        algo_bytes = n_bytes * 1 + n_rec * 9  # 1 B/base + u64 offset + 1 B flag per record, per launch
        achieved = algo_bytes / (k_avg_ms * 1e-3) / 1e9
        job_bytes = ((n_bytes + 9 * n_rec) * world if args.ragged else total * (L + 9)) * n_mates * args.steps
        info = dict(m.filter_info(), **m.filter_mode())
        what = "pairs" if args.paired else "reads"
        out = {
            "metric": "Gbases/s scanned (150bp FASTQ, 10k 31-mers); % HBM roofline at 1/2/4/8 GPUs",
            "value": round(value, 3),
            "unit": "Gbases/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": f"extract{' -2 (paired)' if args.paired else ''} ({'all hit tuples' if emit else 'any-hit flags'}): "
                            f"{total} x {L} bp synthetic {what} in the job, {n_rec} per GPU on rank 0, "
                            f"{len(patterns)} {args.k}-mers ({'all random' if args.random_kmers else 'half sampled from reads, half random'}), "
                            f"Aho-Corasick semantics, 1/{args.plant_every} reads planted",
                "records_total": total, "records_per_gpu": n_rec, "paired": args.paired, "read_len": L,
                "patterns": len(patterns), "k": args.k,
                "sharding": f"contiguous record ranges x{world}" + (", pairs unsplit" if args.paired else ""),
                "counter_reduction": reduce_via, "kernel": m.kernel_name, "filter": info,
                "record_lookup": "offsets array" if (args.ragged or args.with_offsets) else "fixed record length (computed)",
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
                "launches_per_step": n_mates,
                # whole job against N x peak (wall time, all ranks): the multi-GPU roofline figure
                "job_frac_of_n_x_peak": round(job_bytes / dt / 1e9 / (HBM_PEAK_GBS * world), 4),
            },
            "summary": {"hits": int(summ[mk.MK_SUM_HITS]), "records_hit": int(summ[mk.MK_SUM_RECORDS_HIT]),
                        "records": int(summ[mk.MK_SUM_RECORDS]), "bases": int(summ[mk.MK_SUM_BASES]),
                        "filter_candidates": int(summ[mk.MK_SUM_CANDIDATES])},
        }
        if emit and order_s:
            o_ms = float(np.mean(order_s)) * 1e3
            out["roofline"]["order_ms_avg"] = round(o_ms, 4)
            out["roofline"]["order"] = dict(m.order_info(), tuples_per_launch=int(d_nh.item()),
                                            note="mk_order_hits_device: histogram + scatter + LDS sort per bin, inside the timed step")
            # the scan and the ordering together against the same algorithmic bytes
            out["roofline"]["frac_scan_plus_order"] = round(algo_bytes / ((k_avg_ms + o_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        out["ranks"] = per_rank
        if rehearsal:
            out["rehearsal"] = True
            out["rehearsal_note"] = f"{world} ranks share {n_dev} GPU(s): functional check, not a scaling measurement"
        out["roofline"]["traffic"], out["roofline"]["traffic_source"] = measured_traffic(m.kernel_name, n_rec, L, len(patterns), args.plant_every)
        if world == 1 and not args.no_cpu_baseline and not args.paired and not args.ragged:
            out["cpu_baseline"] = cpu_baseline(mk, m, patterns, seed, n_rec, L, args.plant_every, mates[0][2],
                                               args.cpu_seconds)
        default_workload = default_workload_flag(args, options)
        if world == 1 and default_workload and not (args.no_other_configs or args.paired or args.ragged or args.with_offsets):
            try:  # (whatever happens in the side runs, the headline line must reach the driver -- with the failure named in it)
                out["other_configs"] = other_configs(mk, lib, torch, dev, dev_index, m, mates[0], n_rec, L, seed, st)
            except Exception as e:
                out["other_configs"] = f"FAILED: {e!r}"
    if world > 1 and default_workload_flag(args, options) and not (args.no_other_configs or args.paired or args.ragged or args.with_offsets):
        # BASELINE's multi-GPU configurations next to the headline (every rank takes part; rank 0 reports): config 3
        # (paired, pairs unsplit) and config 5 (500 k 21-mers, level-1 filter in global memory), one shard per rank
        del mates, d_keep
        torch.cuda.empty_cache()
        # (these extra runs hold collectives of their own; whatever happens in them, the headline line must reach the driver:
        # if they have not come back after three minutes, rank 0 prints the line without them and the job ends)
        import threading

        def give_up():
            out["other_configs"] = "not finished within 180 s: the headline above stands on its own"
            print(json.dumps(out), flush=True)
            os._exit(0)

        watchdog = threading.Timer(180.0, give_up) if rank == 0 else None
        if watchdog:
            watchdog.daemon = True
            watchdog.start()
        try:
            oc = other_configs_multi(mk, lib, torch, dist if use_dist else None, rehearsal, dev, dev_index, m, rank, world, seed, st, barrier)
        except Exception as e:  # this rank's failure is reported; its peers meet the watchdog
            oc = f"failed on rank {rank}: {e!r}"
        if watchdog:
            watchdog.cancel()
        if rank == 0:
            out["other_configs"] = oc
    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist:
        barrier()
        if reduce_via.startswith("mk_comm"):  # every rank is past its last collective: drop the C ABI's communicator now,
            lib.mk_comm_destroy(m.handle)     # not at interpreter exit
        dist.destroy_process_group()


def default_workload_flag(args, options):
    return (args.records, args.read_len, args.patterns, args.k, args.mode, args.plant_every, args.rc, args.scaling) == \
        (100_000_000, 150, 10_000, 31, "any", 100, False, "weak") and options is None


def other_configs_multi(mk, lib, torch, dist, rehearsal, dev, dev_index, m0, rank, world, seed, st, barrier, steps=5, warmup=2):
    """N > 1: BASELINE's own multi-GPU configurations, weak-scaled like the headline (every rank one 8-GPU-job shard):
    config 3 -- extract paired, 2 x 6.25 M x 150 bp mates per rank, the headline's 10 k 31-mers, a pair is kept if either mate
    hits, pairs never split; config 5 -- 12.5 M x 250 bp per rank, 500 k 21-mers (level-1 filter in global memory).
    Same contract as the headline: barrier + synchronize on both sides, the slowest rank's wall time, aggregate bases over
    all ranks; per-rank kernel times gathered next to it."""
    import numpy as np
    res = []

    def timed(step, m, launches):
        for _ in range(warmup):
            step()
        barrier()
        m.enable_timing(launches * steps)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        barrier()
        dt = time.perf_counter() - t0
        k = m.kernel_times_ms()
        mine = {"rank": rank, "device": dev_index, "kernel_ms_avg": round(float(np.mean(k)), 4), "wall_ms_per_step": round(dt / steps * 1e3, 4)}
        allr = [mine]
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
            allr = [None] * world
            dist.all_gather_object(allr, mine)
        return dt, allr

    def entry(label, m, n_rec, L, launches, dt, allr, extra):
        algo = n_rec * L + 9 * n_rec
        k_ms = float(np.mean([r["kernel_ms_avg"] for r in allr]))
        r = {"workload": label, "n_gpus": world, "scaling": "weak", "kernel": m.kernel_name, "steps": steps, "launches_per_step": launches,
             "ms_per_step": round(dt / steps * 1e3, 4), "value_gbases_per_s": round(launches * n_rec * L * world * steps / dt / 1e9, 1),
             "kernel_ms": round(k_ms, 4), "algorithmic_bytes_per_launch": algo, "frac": round(algo / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
             "job_frac_of_n_x_peak": round(launches * algo * world * steps / dt / 1e9 / (HBM_PEAK_GBS * world), 4), "ranks": allr,
             "traffic": None, "traffic_source": "multi-GPU run: no PMC pass"}
        r.update(extra)
        res.append(r)

    # ---- config 3: this rank's pairs [rank * n3, (rank + 1) * n3) of the job's two mate files
    n3, L3 = 6_250_000, 150
    pair = []
    for f in range(2):
        d_seq = torch.empty(n3 * L3 + 64, dtype=torch.uint8, device=dev)
        d_off = torch.empty(n3 + 1, dtype=torch.int64, device=dev)
        d_flags = torch.empty((n3 + 7) // 4 * 4, dtype=torch.uint8, device=dev)
        mk._check(lib.mk_synth_reads_device_range(m0.handle, seed + 20 + f, rank * n3, n3, L3, 100, d_seq.data_ptr(), d_off.data_ptr(), st))
        pair.append((d_seq, d_off, d_flags))
    d_keep = torch.empty_like(pair[0][2])
    d_nh = torch.zeros(1, dtype=torch.int64, device=dev)
    d_cnt = torch.zeros(len(m0.patterns) + mk.MK_NUM_SUMMARY, dtype=torch.int64, device=dev)
    mk._check(lib.mk_matcher_set_fixed_record_length(m0.handle, L3))
    mk._check(lib.mk_matcher_hint_hit_density(m0.handle, 10))

    def step3():
        for d_seq, d_off, d_flags in pair:
            mk._check(lib.mk_scan_device(m0.handle, d_seq.data_ptr(), n3 * L3, d_off.data_ptr(), n3, mk.MK_MODE_ANY, d_flags.data_ptr(), None, 0,
                                         d_nh.data_ptr(), d_cnt.data_ptr(), st))
        torch.bitwise_or(pair[0][2], pair[1][2], out=d_keep)

    dt, allr = timed(step3, m0, 2)
    kept = torch.tensor([int(d_keep[:n3].sum().item())], dtype=torch.int64, device="cpu" if (rehearsal or dist is None) else dev)
    if dist is not None:
        dist.all_reduce(kept)
    entry(f"config 3: extract paired, 2 x {n3 * world} x {L3} bp mates in the job ({n3} pairs per GPU), {len(m0.patterns)} 31-mers, any-hit flags, "
          "pair kept if either mate hits, pairs unsplit", m0, n3, L3, 2, dt, allr, {"pairs_kept_in_the_job": int(kept.item())})
    del pair, d_keep

    # ---- config 5: this rank's reads of the 100 M x 250 bp job, 500 k 21-mers
    n5, L5 = 12_500_000, 250
    pats = mk.parse_pattern_list(kmer_seq=make_patterns(500_000, 21, seed=13))[:500_000]
    m5 = mk.Matcher(pats, device=dev_index)
    d_seq = torch.empty(n5 * L5 + 64, dtype=torch.uint8, device=dev)
    d_off = torch.empty(n5 + 1, dtype=torch.int64, device=dev)
    d_flags = torch.empty((n5 + 7) // 4 * 4, dtype=torch.uint8, device=dev)
    mk._check(lib.mk_synth_reads_device_range(m5.handle, seed + 7, rank * n5, n5, L5, 100, d_seq.data_ptr(), d_off.data_ptr(), st))
    d_cnt5 = torch.zeros(len(pats) + mk.MK_NUM_SUMMARY, dtype=torch.int64, device=dev)
    mk._check(lib.mk_matcher_set_fixed_record_length(m5.handle, L5))
    mk._check(lib.mk_matcher_hint_hit_density(m5.handle, 10))

    def step5():
        mk._check(lib.mk_scan_device(m5.handle, d_seq.data_ptr(), n5 * L5, d_off.data_ptr(), n5, mk.MK_MODE_ANY, d_flags.data_ptr(), None, 0,
                                     d_nh.data_ptr(), d_cnt5.data_ptr(), st))

    dt, allr = timed(step5, m5, 1)
    entry(f"config 5: extract, {n5 * world} x {L5} bp in the job ({n5} reads per GPU), {len(pats)} 21-mers (level-1 filter in global memory), "
          "any-hit flags", m5, n5, L5, 1, dt, allr, {"filter": dict(m5.filter_info(), **m5.filter_mode())})
    m5.close()
    return res


def other_configs(mk, lib, torch, dev, dev_index, m0, mate0, n_rec0, L0, seed, st, steps=5, warmup=2):
    """Short runs (5 timed steps each) of the other BASELINE configurations that fit one GPU and of the headline
    batch with 10 % / all of the reads hitting, appended to the headline's JSON line: the same step as the
    headline (clear + scan + flag count; hits mode: + the tuples put into the reference's emission order), the
    scan kernel timed by hipEvents, fractions against the same 8 TB/s and the same algorithmic bytes
    (1 B per base + 9 B per record)."""
    import numpy as np
    res = []

    def run(label, m, d_seq, d_off, d_flags, n_rec, L, n_pat, emit, plant_every):
        n_bytes = n_rec * L
        d_nh = torch.zeros(1, dtype=torch.int64, device=dev)
        d_cnt = torch.zeros(n_pat + mk.MK_NUM_SUMMARY, dtype=torch.int64, device=dev)
        cap = max(1 << 20, 2 * n_rec // max(1, plant_every)) if emit else 0
        d_hits = torch.empty(2 * cap, dtype=torch.int64, device=dev) if emit else None
        order_s = []

        def step():
            mk._check(lib.mk_scan_device(m.handle, d_seq.data_ptr(), n_bytes, d_off.data_ptr(), n_rec,
                                         mk.MK_MODE_HITS if emit else mk.MK_MODE_ANY, d_flags.data_ptr(),
                                         d_hits.data_ptr() if emit else None, cap, d_nh.data_ptr(), d_cnt.data_ptr(), st))
            if emit:
                nh = int(d_nh.item())
                if nh > cap:
                    raise RuntimeError(f"{label}: {nh} tuples do not fit the buffer of {cap}")
                t_o = time.perf_counter()
                mk._check(lib.mk_order_hits_device(m.handle, d_hits.data_ptr(), nh, st))
                torch.cuda.synchronize()
                order_s.append(time.perf_counter() - t_o)

        mk._check(lib.mk_matcher_set_fixed_record_length(m.handle, L))
        mk._check(lib.mk_matcher_hint_hit_density(m.handle, 0))
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        w = d_cnt.cpu().numpy()[n_pat:]
        mk._check(lib.mk_matcher_hint_hit_density(m.handle, int(w[mk.MK_SUM_RECORDS_HIT]) * 1000 // max(1, int(w[mk.MK_SUM_RECORDS]))))
        step()  # the flavour the hint selects, warm
        torch.cuda.synchronize()
        d_cnt.zero_()
        del order_s[:]
        m.enable_timing(steps)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        k_ms = float(np.mean(m.kernel_times_ms()))
        algo = n_bytes + 9 * n_rec
        summ = d_cnt.cpu().numpy()[n_pat:]
        info = dict(m.filter_info(), **m.class_info())
        r = {"workload": label, "kernel": m.kernel_name, "filter": info, "steps": steps, "ms_per_step": round(dt / steps * 1e3, 4),
             "value_gbases_per_s": round(n_bytes * steps / dt / 1e9, 1), "kernel_ms": round(k_ms, 4),
             "algorithmic_bytes_per_launch": algo, "frac": round(algo / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
             "records_hit_per_launch": int(summ[mk.MK_SUM_RECORDS_HIT]) // steps, "hits_per_launch": int(summ[mk.MK_SUM_HITS]) // steps,
             "filter_candidates_per_launch": int(summ[mk.MK_SUM_CANDIDATES]) // steps}
        # counter traffic exists for workloads that were profiled under rocprofv3 with this kernel source (profiles/traffic_*.json)
        r["traffic"], r["traffic_source"] = measured_traffic(m.kernel_name, n_rec, L, n_pat, plant_every)
        if r["traffic"] is None and r["traffic_source"] is None:
            r["traffic_source"] = "no committed PMC profile of this workload (tools/profile_gpu.sh writes one per workload it is run on)"
        if emit:
            o_ms = float(np.mean(order_s)) * 1e3
            r["order_ms"] = round(o_ms, 4)
            r["order"] = m.order_info()
            r["frac_scan_plus_order"] = round(algo / ((k_ms + o_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        res.append(r)

    # the headline batch again with 10 % and all of the reads hitting (tag on already extracted reads)
    d_seq, d_off, d_flags = mate0
    for pe, what in ((10, "10 % of the reads hit"), (1, "every read hits")):
        mk._check(lib.mk_synth_reads_device_range(m0.handle, seed, 0, n_rec0, L0, pe, d_seq.data_ptr(), d_off.data_ptr(), st))
        run(f"headline batch, {what}: {n_rec0} x {L0} bp, {len(m0.patterns)} 31-mers, any-hit flags", m0, d_seq, d_off, d_flags,
            n_rec0, L0, len(m0.patterns), False, pe)
        if pe == 1:  # tag / extract -l on already extracted reads: every tuple, in emission order
            run(f"headline batch, {what}: {n_rec0} x {L0} bp, {len(m0.patterns)} 31-mers, every hit tuple in emission order", m0, d_seq,
                d_off, d_flags, n_rec0, L0, len(m0.patterns), True, pe)
    # ---- pattern sets of MIXED lengths (the reference's DFA scans any list at one speed, src/cmd_extract.rs:260-265).
    # Until r04 the whole set ran at the stride its SHORTEST pattern admits; now the short patterns form a class of
    # their own next to the main filter (matcher.cpp: plan_classes) and the q-gram floor follows the pattern count.
    import numpy as np
    plus8 = mk.parse_pattern_list(kmer_seq=list(m0.patterns) + [b"GATTACAG"])
    rng = np.random.default_rng(77)
    mixed = mk.parse_pattern_list(kmer_seq=[p[:int(rng.integers(15, 32))] for p in make_patterns(10_000, 31, seed=5)])
    tiny = mk.parse_pattern_list(kmer_seq=[b"A", b"AA", b"AC", b"AG", b"C", b"CT", b"CTC", b"G", b"GA", b"GAG", b"GT", b"T", b"TC", b"TT"])
    for pats, n_r, what in (
            (plus8, n_rec0, f"{len(m0.patterns)} 31-mers + ONE 8-mer (r03 geometry: S=1, q=8 for the whole set, 71.7 ms)"),
            (mixed, n_rec0, f"{len(mixed)} patterns of 15..31 bases, uniform (r03 geometry: S=2, q=14, 4.15 ms)"),
            (tiny, min(n_rec0, 10_000_000), "the 14 patterns of 1-3 bases of the reference's Aho-Corasick golden (tests/fixtures/extract/log.json): "
                                           "1.5 occurrences per base, bound by the occurrence rate, not by the stream")):
        mm = mk.Matcher(pats, device=dev_index)
        mk._check(lib.mk_synth_reads_device_range(mm.handle, seed, 0, n_r, L0, 100, d_seq.data_ptr(), d_off.data_ptr(), st))
        run(f"mixed lengths: {n_r} x {L0} bp, {what}, any-hit flags", mm, d_seq, d_off, d_flags, n_r, L0, len(pats), False, 100)
        mm.close()
    del d_seq, d_off, d_flags

    def fresh(n_rec, L, n_pat, k, rc, plant_every, s):
        raw = make_patterns(n_pat, k, seed=s)
        pats = mk.parse_pattern_list(kmer_seq=raw)[:n_pat]
        if rc:
            pats = mk.parse_pattern_list(kmer_seq=pats, reverse_complement=True)
        m = mk.Matcher(pats, device=dev_index)
        d_seq = torch.empty(n_rec * L + 64, dtype=torch.uint8, device=dev)
        d_off = torch.empty(n_rec + 1, dtype=torch.int64, device=dev)
        d_flags = torch.empty((n_rec + 7) // 4 * 4, dtype=torch.uint8, device=dev)
        mk._check(lib.mk_synth_reads_device_range(m.handle, seed + 7, 0, n_rec, L, plant_every, d_seq.data_ptr(), d_off.data_ptr(), st))
        return m, d_seq, d_off, d_flags, len(pats)

    # config 3, one GPU's shard of the 8-GPU job: 2 x 6.25 M x 150 bp mates, 10 k 31-mers (the headline's matcher); a pair
    # is kept if either mate hits (src/cmd_extract.rs:600-606)
    n3 = 6_250_000
    pair = []
    for f in range(2):
        d_seq = torch.empty(n3 * L0 + 64, dtype=torch.uint8, device=dev)
        d_off = torch.empty(n3 + 1, dtype=torch.int64, device=dev)
        d_flags = torch.empty((n3 + 7) // 4 * 4, dtype=torch.uint8, device=dev)
        mk._check(lib.mk_synth_reads_device_range(m0.handle, seed + 20 + f, 0, n3, L0, 100, d_seq.data_ptr(), d_off.data_ptr(), st))
        pair.append((d_seq, d_off, d_flags))
    d_keep = torch.empty_like(pair[0][2])
    d_nh3 = torch.zeros(1, dtype=torch.int64, device=dev)
    d_cnt3 = torch.zeros(len(m0.patterns) + mk.MK_NUM_SUMMARY, dtype=torch.int64, device=dev)
    mk._check(lib.mk_matcher_set_fixed_record_length(m0.handle, L0))
    mk._check(lib.mk_matcher_hint_hit_density(m0.handle, 10))

    def step3():
        for d_seq, d_off, d_flags in pair:
            mk._check(lib.mk_scan_device(m0.handle, d_seq.data_ptr(), n3 * L0, d_off.data_ptr(), n3, mk.MK_MODE_ANY, d_flags.data_ptr(), None, 0,
                                         d_nh3.data_ptr(), d_cnt3.data_ptr(), st))
        torch.bitwise_or(pair[0][2], pair[1][2], out=d_keep)

    for _ in range(warmup):
        step3()
    torch.cuda.synchronize()
    m0.enable_timing(2 * steps)
    t0 = time.perf_counter()
    for _ in range(steps):
        step3()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    k_ms = float(np.mean(m0.kernel_times_ms()))
    algo = n3 * L0 + 9 * n3
    res.append({"workload": f"config 3, one GPU's shard: extract paired, 2 x {n3} x {L0} bp mates, {len(m0.patterns)} 31-mers, any-hit flags, pair kept if either mate hits",
                "kernel": m0.kernel_name, "steps": steps, "launches_per_step": 2, "ms_per_step": round(dt / steps * 1e3, 4),
                "value_gbases_per_s": round(2 * n3 * L0 * steps / dt / 1e9, 1), "kernel_ms": round(k_ms, 4),
                "algorithmic_bytes_per_launch": algo, "frac": round(algo / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "pairs_kept": int(d_keep[:n3].sum().item())})
    res[-1]["traffic"], res[-1]["traffic_source"] = measured_traffic(m0.kernel_name, n3, L0, len(m0.patterns), 100)
    if res[-1]["traffic"] is None and not res[-1]["traffic_source"]:
        res[-1]["traffic_source"] = "no committed PMC profile of this workload (tools/profile_gpu.sh writes one per workload it is run on)"
    del pair, d_keep

    m, a, b, c, n_pat = fresh(10_000_000, 150, 1024, 31, True, 100, 11)
    run(f"config 2 shape: extract, 10 M x 150 bp, 1 024 31-mers + RC ({n_pat} patterns), any-hit flags", m, a, b, c, 10_000_000, 150, n_pat, False, 100)
    del m, a, b, c
    m, a, b, c, n_pat = fresh(20_000_000, 150, 10_000, 31, False, 100, 12)
    run("config 4 shape: tag, 20 M x 150 bp records, 10 k 31-mers, every hit tuple in emission order", m, a, b, c, 20_000_000, 150, n_pat, True, 100)
    del m, a, b, c
    m, a, b, c, n_pat = fresh(12_500_000, 250, 500_000, 21, False, 100, 13)
    run("config 5, one GPU's shard: extract, 12.5 M x 250 bp, 500 k 21-mers (filter in global memory), any-hit flags", m, a, b, c,
        12_500_000, 250, n_pat, False, 100)
    del m, a, b, c
    try:
        res += codec_configs(mk)
    except Exception as e:  # (a failed check of the codec entries is reported as one; the scan entries above stand)
        res.append({"workload": "BGZF codec / window entries", "error": repr(e)})
    return res


def _fastq_binned(n, L=150, seed=11):
    """FASTQ text with Illumina-style binned qualities in runs (a position keeps its predecessor's bin with p = 0.92, the lower
    bins open up along the read): what a sequencer's bgzip'ed output looks like to a DEFLATE coder, unlike constant qualities"""
    import numpy as np
    rng = np.random.default_rng(seed)
    bins = np.frombuffer(b"FFF:,#", dtype=np.uint8)
    q = np.empty((n, L), dtype=np.uint8)
    cur = rng.integers(0, 2, size=n).astype(np.uint8)
    for j in range(L):
        change = rng.random(n) < 0.08
        nxt = rng.integers(0, 3 + (3 * j) // L, size=n).astype(np.uint8)
        cur = np.where(change, nxt, cur)
        q[:, j] = bins[cur]
    H = 13
    rec = np.empty((n, H + L + 3 + L + 1), dtype=np.uint8)
    rec[:, :H] = np.array([f"@r{i:010d}\n" for i in range(n)], dtype="S13").view(np.uint8).reshape(n, H)
    rec[:, H:H + L] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n, L))]
    rec[:, H + L:H + L + 3] = np.frombuffer(b"\n+\n", dtype=np.uint8)
    rec[:, H + L + 3:H + 2 * L + 3] = q
    rec[:, -1] = ord("\n")
    return rec.tobytes()


def _zlib_bgzf(raw, level=6):
    """BGZF members as htslib / bgzip write them: zlib at `level`, 65 280 bytes of text each (16 host threads: zlib releases the GIL)"""
    import zlib
    from concurrent.futures import ThreadPoolExecutor

    def member(b):
        chunk = raw[b:b + 0xff00]
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        c = co.compress(chunk) + co.flush()
        return (bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0]) + (len(c) + 25).to_bytes(2, "little") + c +
                zlib.crc32(chunk).to_bytes(4, "little") + len(chunk).to_bytes(4, "little"))
    with ThreadPoolExecutor(16) as ex:
        return b"".join(ex.map(member, range(0, len(raw), 0xff00)))


def codec_configs(mk, megabytes=1024, reps=3):
    """The BGZF codec either side of `tag` and in front of `extract` (include/merkurio_hip.h, merkurio_amd/csrc/codec/), through the C
    ABI with host buffers in and out.  r05: the inflater is measured on members a REAL writer made -- zlib level 6, what htslib and
    bgzip produce -- of 1 GiB of BAM-shaped records (tests/textio.py: bam_like, 4-bin qualities) and of 1 GiB of FASTQ with binned
    qualities in runs, and on 64 MB of the same (a small call: the wave-per-member kernel); the members the device's own deflate
    writes (longer matches, fewer tokens) stay beside them, labelled.  `kernel_ms` = the device part of a call (hipEvents around its
    kernels), best of `reps`; `ms_per_call` = the C call alone, from pageable host memory into a fresh, untouched output buffer
    (upload_ms / download_ms: its two host <-> device legs through the handle's page-locked staging buffers).  Neither direction is
    bound by HBM or MFMA -- latency-bound serial decoders / parsers per member -- so `frac` against the HBM peak is for scale only.
    Every round trip is checked in the run."""
    import zlib
    from textio import bam_like
    codec = mk.Codec()
    out = []
    n_bytes = megabytes << 20
    unit = bam_like(200000, seed=21)
    bam = (unit * (n_bytes // len(unit) + 1))[:n_bytes]
    # what the reference runs on its reader / writer threads, on one host core, on a 16 MB sample of the same text (zlib: the
    # checker of the codec tests; the reference's flate2 backend is of the same class)
    sample = bam[:256 * 65280]
    t0 = time.perf_counter()
    zs = []
    for i in range(0, len(sample), 65280):
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        zs.append(co.compress(sample[i:i + 65280]) + co.flush())
    z_def = len(sample) / (time.perf_counter() - t0) / 1e6
    t0 = time.perf_counter()
    for z in zs:
        zlib.decompress(z, -15)
    z_inf = len(sample) / (time.perf_counter() - t0) / 1e6
    z_ratio = len(sample) / (sum(map(len, zs)) + 26 * len(zs))

    def timed(fn):
        best = None
        for _ in range(reps):
            res = fn()
            up, dev_ms, down = codec.times()
            if best is None or dev_ms < best[1]:
                best = (codec.last_call_s, dev_ms, up, down)
        return res, best

    def entry(workload, kernel, n_text, n_blob, best, bound, extra):
        e = {"workload": workload, "kernel": kernel, "kernel_ms": round(best[1], 2), "ms_per_call": round(best[0] * 1e3, 1),
             "upload_ms": round(best[2], 1), "download_ms": round(best[3], 1), "legs_share_of_call": round((best[1] + best[2] + best[3]) / (best[0] * 1e3), 3),
             "text_gb_per_s_kernels": round(n_text / best[1] / 1e6, 1), "text_gb_per_s_call": round(n_text / best[0] / 1e9, 2), "bound": bound,
             "frac": round((n_text + n_blob) / (best[1] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
             "traffic_source": "not an HBM-bound kernel: no PMC traffic profile"}
        e.update(extra)
        out.append(e)

    try:
        blob, best = timed(lambda: codec.deflate(bam))
        entry(f"BGZF deflate (tag's BAM output): {len(bam) / 1e6:.0f} MB of BAM-shaped records -> {(len(bam) + 65279) // 65280} members",
              "mk_bgzf_crc_kernel + mk_bgzf_deflate_kernel + mk_bgzf_pack_kernel", len(bam), len(blob), best,
              "latency (serial parse per member; not HBM, not MFMA)",
              {"compression_ratio": round(len(bam) / len(blob), 2),
               "cpu_baseline": {"value": round(z_def / 1e3, 4), "unit": "GB/s of text", "cores": 1, "kind": "zlib level 6, 65 280-byte members",
                                "compression_ratio": round(z_ratio, 2), "sample": "the first 16.7 MB of the same text"}})
        inflate_kernels = "mk_bgzf_inflate_wave_kernel<4096 / 2048> (a wave per member, its recent 4 / 2 KiB in LDS: calls of up to ~2 000 / ~24 000 members) or mk_bgzf_inflate_kernel (a lane per member) + mk_bgzf_crc_check_kernel"
        cpu_inf = {"value": round(z_inf / 1e3, 4), "unit": "GB/s of text", "cores": 1, "kind": "zlib inflate of its own level 6 members",
                   "sample": "the first 16.7 MB of the BAM text"}
        text, best = timed(lambda: codec.inflate(blob))
        entry(f"BGZF inflate, members the DEVICE wrote (long matches, few tokens): {(len(bam) + 65279) // 65280} members -> {len(bam) / 1e6:.0f} MB of BAM "
              "records, CRC-32 checked", inflate_kernels, len(bam), len(blob), best, "latency (serial decode per member; not HBM, not MFMA)",
              {"round_trip_equal": bool(text == bam), "cpu_baseline": cpu_inf})
        if text != bam:
            raise RuntimeError("BGZF round trip on the device does not reproduce its input")
        del text, blob
        fq = _fastq_binned(n_bytes // 317 + 1)[:n_bytes]
        for label, raw in (("BAM records, 4-bin qualities", bam), ("FASTQ, binned qualities in runs", fq)):
            zb = _zlib_bgzf(raw)
            for size_mb in (megabytes, 64):
                tab, _, _ = mk.bgzf_members(zb)
                import numpy as np
                cum = np.cumsum(tab["isize"].astype(np.int64))
                k = min(len(tab), int(np.searchsorted(cum, size_mb << 20)) + 1)
                end = int(tab["data_off"][k - 1]) + int(tab["data_len"][k - 1]) + 8
                part, want = zb[:end], raw[:int(cum[k - 1])]
                text, best = timed(lambda: codec.inflate(part))
                ok = text == want
                entry(f"BGZF inflate, zlib LEVEL-6 members (what htslib / bgzip write) of {label}: {k} members -> {len(want) / 1e6:.0f} MB, CRC-32 checked",
                      inflate_kernels, len(want), len(part), best, "latency (serial decode per member; not HBM, not MFMA)",
                      {"compression_ratio": round(len(want) / len(part), 2), "round_trip_equal": bool(ok), "cpu_baseline": cpu_inf})
                if not ok:
                    raise RuntimeError("the device does not inflate zlib's members to their text")
                del text
            del zb
        del fq, bam
        out.append(bgzf_window_config(mk, codec, reps))
        out.append(bam_window_config(mk, codec, reps))
        out.append(gunzip_config(mk, reps))
    finally:
        codec.close()
    return out


def gunzip_config(mk, reps, n_reads=1_500_000):
    """ONE gzip member (what plain `gzip` writes: one DEFLATE stream, no member boundaries to split at) inflated in parallel pieces on
    the device: mk_gzip_inflate_device -- block starts found by trying every bit position, a wave per piece decoding into
    16-bit symbols with place-holders for the unknown 32 KiB in front, contexts by composing maps, CRC-32 / ISIZE checked.  Checked in
    the run against the text that went in; zlib on one host thread inflates the same stream beside it."""
    import zlib
    data = _fastq_binned(n_reads)
    co = zlib.compressobj(6, zlib.DEFLATED, 31)
    gz = co.compress(data) + co.flush()
    t0 = time.perf_counter()
    ok = zlib.decompress(gz, 31) == data
    t_z = time.perf_counter() - t0
    codec = mk.Codec()
    try:
        best = None
        for _ in range(reps + 1):
            text = codec.gunzip(gz)
            if text is None:
                raise RuntimeError(f"mk_gzip_inflate_device did not take a gzip-written FASTQ: {codec.gzip_info}")
            ok = ok and text == data
            del text
            if best is None or codec.last_call_s < best[0]:
                best = (codec.last_call_s, codec.gzip_info)
    finally:
        codec.close()
    if not ok:
        raise RuntimeError("mk_gzip_inflate_device: the text differs from what was compressed")
    pieces, ms = best[1]
    return {"workload": f"one-member gzip (gzip -6): {n_reads} x 150 bp FASTQ reads = {len(data) / 1e6:.0f} MB of text in ONE DEFLATE stream of {len(gz) / 1e6:.0f} MB, "
                        f"inflated in {pieces} parallel pieces; the text stays on the device (extract's windows are ranges of it)",
            "kernel": "mk_gzip_find_kernel + mk_gzip_segments_wave_kernel + mk_gzip_maps_* + mk_gzip_translate_kernel + mk_bgzf_crc_kernel",
            "ms_per_call": round(best[0] * 1e3, 1), "text_gb_per_s_call": round(len(data) / best[0] / 1e9, 2),
            "phases_ms": dict(zip(("upload", "block_search", "pieces", "contexts_and_text", "crc"), ms)), "pieces": pieces, "round_trip_equal": True,
            "cpu_baseline": {"value": round(len(data) / t_z / 1e9, 3), "unit": "GB/s of text", "cores": 1, "kind": "zlib inflate of the same stream", "sample": "the whole stream"},
            "bound": "the CUs' scalar units (wave-uniform decode); not HBM, not MFMA", "kernel_ms": None, "frac": None, "traffic": None,
            "traffic_source": "a host-buffer call, not a kernel: end-to-end figure"}


def bam_window_config(mk, codec, reps, n_rec=3_000_000, L=150, n_pat=10_000):
    """One window of a BAM through mk_tag_bam_window (BASELINE config 4's shape: 150-base records, 10 000 31-mers, km tag +
    -m): zlib level-6 members in, BGZF members of the tagged kept records out; inflate, record chain, un-nibbling, scan, pattern sets,
    tag append and deflate on the device.  `ms_per_call` is the whole C call (host buffers in and out); `phases_ms` the library's own
    split.  Checked in the run: the members inflate (zlib) to the kept records with `km:Z:<pattern>` appended, and the records kept are
    the planted ones."""
    import ctypes as C
    import struct
    import zlib
    import numpy as np
    lib = mk.load()
    rng = np.random.default_rng(9)
    code = np.zeros(256, dtype=np.uint8)
    code[list(b"ACGT")] = [1, 2, 4, 8]
    pats = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n_pat, 31))]
    nib = np.array([1, 2, 4, 8], dtype=np.uint8)[rng.integers(0, 4, size=(n_rec, L))]
    planted = np.arange(0, n_rec, 100)
    nib[planted, 8:39] = code[pats[planted % n_pat]]
    W = 36 + 12 + 4 + L // 2 + L + 15
    rec = np.zeros((n_rec, W), dtype=np.uint8)
    rec[:, 0:4] = np.frombuffer(struct.pack("<I", W - 4), dtype=np.uint8)
    rec[:, 8:12] = ((np.arange(n_rec, dtype=np.uint32) * 37) % 2000000).view(np.uint8).reshape(n_rec, 4)
    rec[:, 12], rec[:, 13], rec[:, 16] = 12, 60, 1
    rec[:, 20:24] = np.frombuffer(struct.pack("<I", L), dtype=np.uint8)
    rec[:, 24:28] = 255
    rec[:, 28:32] = 255
    rec[:, 36:48] = np.array([b"r%010d\0" % i for i in range(n_rec)], dtype="S12").view(np.uint8).reshape(n_rec, 12)
    rec[:, 48:52] = np.frombuffer(struct.pack("<I", L << 4), dtype=np.uint8)
    rec[:, 52:52 + L // 2] = nib[:, 0::2] << 4 | nib[:, 1::2]
    q0 = 52 + L // 2
    rec[:, q0:q0 + L] = np.array([2, 12, 23, 37], dtype=np.uint8)[rng.choice(4, size=(n_rec, L), p=[0.02, 0.05, 0.13, 0.8])]
    rec[:, q0 + L:] = np.frombuffer(b"NMC\0ASC\x96XSZabc\0", dtype=np.uint8)
    text = rec.tobytes()
    first_planted = rec[0].tobytes()
    del rec, nib
    blob = _zlib_bgzf(text)
    mem, used, total = mk.bgzf_members(blob)
    plist = mk.parse_pattern_list(kmer_seq=[p.tobytes() for p in pats])
    m = mk.Matcher(plist, device=0)
    bb = np.frombuffer(blob, dtype=np.uint8)
    w = mk.BamWindow()
    tail, out = np.zeros(1 << 20, dtype=np.uint8), np.zeros(len(text) // 16 + (1 << 20), dtype=np.uint8)
    w.bgzf, w.n_bgzf, w.members, w.n_members = bb.ctypes.data, bb.size, mem.ctypes.data, len(mem)
    w.last, w.filter_matching, w.tag[0], w.tag[1] = 1, 1, ord("k"), ord("m")
    w.tail, w.tail_cap, w.out, w.out_cap = tail.ctypes.data, tail.size, out.ctypes.data, out.size
    status = C.c_uint32()

    def call():
        cnt = mk.Counters()
        mk._check(lib.mk_tag_bam_window(m.handle, codec._h, C.byref(w), 0, C.byref(cnt), None, C.byref(status)))

    call()
    ts, phases = [], None
    for _ in range(reps):
        t0 = time.perf_counter()
        call()
        ts.append(time.perf_counter() - t0)
        if ts[-1] == min(ts):
            phases = [round(float(x), 2) for x in w.ms]
    got = b""
    d = zlib.decompressobj(31)
    buf = out[:w.out_len].tobytes()
    while buf:  # (concatenated members)
        got += d.decompress(buf)
        buf = d.unused_data
        d = zlib.decompressobj(31)
    p0 = pats[0].tobytes()
    want0 = struct.pack("<I", W - 4 + 3 + 31 + 1) + first_planted[4:] + b"kmZ" + p0 + b"\0"
    ok = bool(status.value == 0 and w.n_rec == n_rec and w.n_kept >= len(planted) and w.n_kept < len(planted) + 64 and got[:len(want0)] == want0 and
              len(got) == w.out_text_bytes)
    if not ok:
        raise RuntimeError(f"mk_tag_bam_window: the tagged records do not check out (status {status.value}, {w.n_rec} records, {w.n_kept} kept, "
                           f"{len(got)} of {w.out_text_bytes} bytes)")
    return {"workload": f"BAM window (config-4 shape): {n_rec} x {L}-base records = {len(text) / 1e6:.0f} MB of BAM text in {len(mem)} zlib level-6 members "
                        f"({len(blob) / 1e6:.0f} MB), {n_pat} 31-mers, km tag + -m; mk_tag_bam_window: members in, members of the {int(w.n_kept)} kept records out",
            "kernel": "inflate + mk_bam_find / walk / unpack + " + m.kernel_name + " + order / sets + mk_bam_taglen / emit + deflate",
            "ms_per_call": round(min(ts) * 1e3, 1), "text_gb_per_s_call": round(len(text) / min(ts) / 1e9, 2),
            "gbases_per_s_call": round(n_rec * L / min(ts) / 1e9, 2),
            "phases_ms": dict(zip(("upload", "inflate", "record_index", "unpack_scan_sets", "tag_pack", "deflate", "download", "of_these_growing_buffers"), phases)),
            "records_kept": int(w.n_kept), "tagged_records_check": ok, "bound": "latency of the inflate launch + PCIe of the members",
            "kernel_ms": None, "frac": None, "traffic": None, "traffic_source": "a host-buffer call, not a kernel: end-to-end figure"}


def bgzf_window_config(mk, codec, reps, n_reads=3_000_000, L=150, n_pat=10_000):
    """One window of a bgzip'ed FASTQ through mk_extract_fastq_bgzf: the members go up as they are, are inflated
    straight into the ingest buffer, indexed, gathered, scanned; the unfinished tail and the kept records come back.
    `ms_per_call` is the whole call (host buffers in and out), against mk_extract_fastq_text on the same text (the text
    uploaded instead); the flags of the two are compared in the run."""
    import ctypes as C
    import numpy as np
    lib = mk.load()
    rng = np.random.default_rng(5)
    pats = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n_pat, 31))]
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n_reads, L))]
    for i in range(0, n_reads, 100):
        bases[i, 7:38] = pats[i % n_pat]
    H = 13
    rec = np.empty((n_reads, H + L + 3 + L + 1), dtype=np.uint8)
    rec[:, :H] = np.array([b"@r%010d\n" % i for i in range(n_reads)], dtype="S13").view(np.uint8).reshape(n_reads, H)
    rec[:, H:H + L] = bases
    rec[:, H + L:H + L + 3] = np.frombuffer(b"\n+\n", dtype=np.uint8)
    rec[:, H + L + 3:H + 2 * L + 3] = ord("I")
    rec[:, -1] = ord("\n")
    text = rec.tobytes()
    del rec, bases
    blob = codec.deflate(text)
    mem, used, total = mk.bgzf_members(blob)
    m = mk.Matcher([p.tobytes() for p in pats], device=0)
    bb = np.frombuffer(blob, dtype=np.uint8)
    tb = np.frombuffer(text, dtype=np.uint8)
    cap = n_reads + 16
    rec_start, keep, keep2 = np.zeros(cap + 1, dtype=np.uint64), np.zeros(cap, dtype=np.uint8), np.zeros(cap, dtype=np.uint8)
    tail, kept = np.zeros(1 << 20, dtype=np.uint8), np.zeros(len(text) // 8, dtype=np.uint8)
    counts = np.zeros(n_pat, dtype=np.uint32)
    n_rec, n_rows, status = C.c_uint64(), C.c_uint64(), C.c_uint32()

    def bgzf_call():
        io = mk.WindowText()
        io.tail, io.tail_cap, io.kept, io.kept_cap = tail.ctypes.data, tail.size, kept.ctypes.data, kept.size
        cnt = mk.Counters()
        mk._check(lib.mk_extract_fastq_bgzf(m.handle, codec._h, None, 0, bb.ctypes.data, bb.size, mem.ctypes.data, len(mem), 1, C.byref(io), 0, 0, cap,
                                            C.byref(n_rec), rec_start.ctypes.data, keep.ctypes.data, None, 0, C.byref(n_rows), C.byref(cnt),
                                            counts.ctypes.data, C.byref(status)))
        return io, cnt

    def text_call():
        cnt = mk.Counters()
        mk._check(lib.mk_extract_fastq_text(m.handle, tb.ctypes.data, tb.size, 0, 0, cap, C.byref(n_rec), rec_start.ctypes.data, keep2.ctypes.data, None, 0,
                                            C.byref(n_rows), C.byref(cnt), counts.ctypes.data, C.byref(status)))
        return cnt

    best = {}
    for name, fn in (("bgzf", bgzf_call), ("text", text_call)):
        fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            r = fn()
            ts.append(time.perf_counter() - t0)
        best[name] = (min(ts), r)
    io, cnt = best["bgzf"][1]
    same = bool(status.value == 0 and np.array_equal(keep[:n_reads], keep2[:n_reads]) and n_rec.value == n_reads)
    if not same:
        raise RuntimeError("mk_extract_fastq_bgzf and mk_extract_fastq_text disagree on the same reads")
    return {"workload": f"bgzip'ed FASTQ window: {n_reads} x {L} bp reads = {len(text) / 1e6:.0f} MB of text in {len(mem)} members ({len(blob) / 1e6:.0f} MB), "
                        f"{n_pat} 31-mers, any-hit flags; mk_extract_fastq_bgzf, tail + kept records back",
            "kernel": "mk_bgzf_inflate_kernel + ingest + " + m.kernel_name, "ms_per_call": round(best["bgzf"][0] * 1e3, 1),
            "text_gb_per_s_call": round(len(text) / best["bgzf"][0] / 1e9, 2), "gbases_per_s_call": round(n_reads * L / best["bgzf"][0] / 1e9, 2),
            "ms_per_call_text_entry": round(best["text"][0] * 1e3, 1), "records_kept": int(keep[:n_reads].sum()), "kept_bytes_back": int(io.n_kept_bytes),
            "flags_equal_to_text_entry": same, "bound": "latency of the inflate launch + PCIe of the members (a fifth of the text)",
            "kernel_ms": None, "frac": None, "traffic": None, "traffic_source": "a host-buffer call, not a kernel: end-to-end figure"}


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
    return min(n, 16)  # a 1-GPU box's CPU share is 16


def cpu_baseline(mk, m, patterns, seed, n_rec, L, plant_every, d_flags, target_s):
    """The reference's CPU path for this workload, restated (oracle, kind "port"): dense
    Aho-Corasick DFA, first-hit break per record (src/cmd_extract.rs:332-335), 1 thread --
    the reference matcher is single-threaded.  Timed on a bounded sample (the first M reads
    of rank 0's shard); the GPU flags for the same reads are checked against it."""
    import numpy as np
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

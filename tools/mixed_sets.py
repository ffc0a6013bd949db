#!/usr/bin/env python3
"""mixed_sets.py -- scan-kernel time of pattern sets whose lengths differ (the K3 row: the reference's DFA scans any
list at one speed, src/cmd_extract.rs:260-265) and of the headline set at every forced stride (calibration of the
geometry cost model in matcher.cpp).

    python tools/mixed_sets.py [--records N] [--steps K] [--only NAME,...] [--single-class]

Sets (synthetic reads of bench.py's generator, 1 % of them planted with a pattern of the set):
  headline      10 000 31-mers
  plus8         10 000 31-mers + one 8-mer
  len15_31      10 000 patterns of lengths 15..31 (uniform)
  short1_3      the 14 patterns of tests/fixtures/extract/log.json (1-3 bytes) on --short-records reads
  s16 .. s1     the headline set at forced stride 16, 8, 4, 2, 1
Prints one line per set: kernel, classes, ms per launch, fraction of the 8 TB/s roofline, candidates, hits.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

LOG_JSON_SET = [b"A", b"AA", b"AC", b"AG", b"C", b"CT", b"CTC", b"G", b"GA", b"GAG", b"GT", b"T", b"TC", b"TT"]


def pattern_sets(mk, bench):
    base = mk.parse_pattern_list(kmer_seq=bench.make_patterns(10_000, 31))[:10_000]
    import numpy as np
    rng = np.random.default_rng(77)
    mixed = []
    for p in bench.make_patterns(10_000, 31, seed=5):
        mixed.append(p[:int(rng.integers(15, 32))])
    return {
        "headline": base,
        "plus8": mk.parse_pattern_list(kmer_seq=base + [b"GATTACAG"]),
        "len15_31": mk.parse_pattern_list(kmer_seq=mixed),
        "short1_3": mk.parse_pattern_list(kmer_seq=LOG_JSON_SET),
    }


def measure(mk, lib, torch, patterns, n_rec, L, steps, options=None, plant_every=100, seed=0x4D65724B7572696F, mode_hits=False):
    import numpy as np
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    m = mk.Matcher(patterns, device=0, options=options)
    n_bytes = n_rec * L
    d_seq = torch.empty(n_bytes + 64, dtype=torch.uint8, device=dev)
    d_off = torch.empty(n_rec + 1, dtype=torch.int64, device=dev)
    d_flags = torch.empty((n_rec + 7) // 4 * 4, dtype=torch.uint8, device=dev)
    d_nh = torch.zeros(1, dtype=torch.int64, device=dev)
    d_cnt = torch.zeros(len(patterns) + mk.MK_NUM_SUMMARY, dtype=torch.int64, device=dev)
    mk._check(lib.mk_synth_reads_device_range(m.handle, seed, 0, n_rec, L, plant_every, d_seq.data_ptr(), d_off.data_ptr(), st))
    mk._check(lib.mk_matcher_set_fixed_record_length(m.handle, L))

    def step():
        mk._check(lib.mk_scan_device(m.handle, d_seq.data_ptr(), n_bytes, d_off.data_ptr(), n_rec, mk.MK_MODE_ANY, d_flags.data_ptr(),
                                     None, 0, d_nh.data_ptr(), d_cnt.data_ptr(), st))

    step()
    torch.cuda.synchronize()
    w = d_cnt.cpu().numpy()[len(patterns):]
    mk._check(lib.mk_matcher_hint_hit_density(m.handle, int(w[mk.MK_SUM_RECORDS_HIT]) * 1000 // max(1, int(w[mk.MK_SUM_RECORDS]))))
    step()
    torch.cuda.synchronize()
    d_cnt.zero_()
    m.enable_timing(steps)
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    k_ms = float(np.mean(m.kernel_times_ms()))
    summ = d_cnt.cpu().numpy()[len(patterns):]
    algo = n_bytes + 9 * n_rec
    info = dict(m.filter_info())
    if hasattr(m, "class_info"):
        info["classes"] = m.class_info()
    res = {"kernel": m.kernel_name, "filter": info, "kernel_ms": round(k_ms, 4), "frac": round(algo / (k_ms * 1e-3) / 8e12, 4),
           "gbases_per_s": round(n_bytes / (k_ms * 1e-3) / 1e9, 1), "candidates": int(summ[mk.MK_SUM_CANDIDATES]) // steps,
           "hits": int(summ[mk.MK_SUM_HITS]) // steps, "records_hit": int(summ[mk.MK_SUM_RECORDS_HIT]) // steps,
           "records": n_rec, "read_len": L, "patterns": len(patterns)}
    m.close()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", type=int, default=100_000_000)
    ap.add_argument("--short-records", type=int, default=10_000_000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--only", default="")
    ap.add_argument("--single-class", action="store_true", help="force one length class (the round-3 geometry)")
    args = ap.parse_args()
    import torch
    import bench
    from merkurio_amd import native as mk
    lib = mk.load()
    sets = pattern_sets(mk, bench)
    names = ["headline", "plus8", "len15_31", "short1_3", "s16", "s8", "s4", "s2", "s1"]
    # calibration of the short class's cost model: the same set at other strides / table kinds (only with --only)
    variants = {"plus8_s2bit": dict(length_classes=2, force_stride2=2), "plus8_s2byte": dict(length_classes=2, force_stride2=2, force_q2=6),
                "plus8_s1bit": dict(length_classes=2, force_stride2=1), "plus8_s1byte": dict(length_classes=2, force_stride2=1, force_q2=6),
                "plus8_s4q4": dict(length_classes=2, force_stride2=4, force_q2=4)}
    if args.only:
        names += list(variants)
    if args.only:
        names = [n for n in names if n in args.only.split(",")]
    if args.only and "c5" in args.only:
        # config 5's pattern set (500 k 21-mers, main filter in global memory) with and without one 8-mer; the
        # one-class geometry of the latter (S = 1, q = 8: every text position is a candidate) on a tenth of the shard
        c5 = mk.parse_pattern_list(kmer_seq=bench.make_patterns(500_000, 21, seed=13))[:500_000]
        c5p = mk.parse_pattern_list(kmer_seq=c5 + [b"GATTACAG"])
        for name, pats, n_rec, opts in (("c5", c5, 12_500_000, None), ("c5plus8", c5p, 12_500_000, None),
                                        ("c5plus8_one_class", c5p, 1_250_000, {"force_single_class": True})):
            if name in args.only.split(","):
                print(name, json.dumps(measure(mk, lib, torch, pats, n_rec, 250, args.steps, options=opts)), flush=True)
    if args.only and "len12" in args.only:
        # 10 000 12-mers at forced strides 1, 2, 4 (q = 12, 11, 9): where the cost model's stride crosses over
        k12 = mk.parse_pattern_list(kmer_seq=bench.make_patterns(10_000, 12, seed=21))[:10_000]
        for st in (0, 1, 2, 4):
            print(f"len12_s{st}", json.dumps(measure(mk, lib, torch, k12, args.records, args.read_len, args.steps,
                                                     options={"force_stride": st} if st else None)), flush=True)
        k9 = mk.parse_pattern_list(kmer_seq=bench.make_patterns(5, 9, seed=22))[:5]
        for st in (0, 2, 4):
            print(f"five9_s{st}", json.dumps(measure(mk, lib, torch, k9, args.records, args.read_len, args.steps,
                                                     options={"force_stride": st} if st else None)), flush=True)
    for name in names:
        opts = {}
        if name.startswith("s") and name[1:].isdigit():
            pats, opts = sets["headline"], {"force_stride": int(name[1:])}
        elif name in variants:
            pats, opts = sets["plus8"], dict(variants[name])
        else:
            pats = sets[name]
        if args.single_class and "force_single_class" in mk.MatcherOptions.__init__.__code__.co_varnames:
            opts["force_single_class"] = True
        n_rec = args.short_records if name == "short1_3" else args.records
        r = measure(mk, lib, torch, pats, n_rec, args.read_len, args.steps, options=opts or None)
        print(name, json.dumps(r), flush=True)


if __name__ == "__main__":
    main()

"""BASELINE.json configs as parity cases (down-scaled copies checked against the oracle) and
full-size, size-independent properties of the device path (run on the GPU box, `-m gpu`).

  C2 extract, 10 M x 150 bp, 1024 31-mers + rev-comp          -> x1/1000 vs oracle
  C3 extract paired, 2 x 50 M x 150 bp, 10 k 31-mers           -> x1/1000 vs oracle
  C4 tag, 20 M records, 10 k 31-mers, km tag + -m filter        -> x1/1000 vs oracle
  C5 extract, 100 M x 250 bp, 500 k 21-mers (set > LDS)         -> full pattern set, 20 k reads vs oracle
  headline 100 M x 150 bp, 10 k 31-mers                         -> full size: properties + oracle on slices
"""
import random

import numpy as np
import pytest

import oracle_binding as ob

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mk():
    from merkurio_amd import native
    native.load()
    if native.device_count() < 1:
        pytest.fail("no HIP device visible")
    return native


def _kmers(n, k, seed):
    rng = np.random.default_rng(seed)
    arr = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n, k))]
    return [arr[i].tobytes() for i in range(n)]


def _reads(n, L, seed, plant=None, every=50):
    rng = np.random.default_rng(seed)
    arr = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n, L))].copy()
    if plant:
        rnd = random.Random(seed)
        for i in range(0, n, every):
            p = np.frombuffer(rnd.choice(plant), dtype=np.uint8)
            o = rnd.randrange(0, L - len(p) + 1)
            arr[i, o:o + len(p)] = p
    return [arr[i].tobytes() for i in range(n)]


def test_config2_scaled(mk):
    patterns = mk.parse_pattern_list(kmer_seq=_kmers(1024, 31, 2), reverse_complement=True)
    recs = _reads(10_000, 150, 22, plant=patterns)
    m = mk.Matcher(patterns)
    assert m.use_ac and m.filter_info()["stride"] == 16
    om = ob.Matcher(patterns, True, 0, False)
    for logging in (True, False):
        for invert in (False, True):
            assert m.extract_single(recs, logging=logging, invert=invert) == \
                ob.extract_single(om, recs, logging=logging, invert=invert)


def test_config3_scaled_paired(mk):
    patterns = mk.parse_pattern_list(kmer_seq=_kmers(10_000, 31, 3))
    r1 = _reads(50_000, 150, 31, plant=patterns, every=97)
    r2 = _reads(50_000, 150, 32, plant=patterns, every=89)
    m = mk.Matcher(patterns)
    assert m.use_ac and m.filter_info() == {"q_gram": 24, "stride": 8, "entries": 80000, "table_bytes": 2097152}
    om = ob.Matcher(patterns, True, 0, False)
    for logging in (True, False):
        assert m.extract_paired(r1, r2, logging=logging) == ob.extract_paired(om, r1, r2, logging=logging)


def test_config4_scaled_tag(mk):
    patterns = mk.parse_pattern_list(kmer_seq=_kmers(10_000, 31, 4))
    recs = _reads(20_000, 150, 41, plant=patterns, every=20)
    m = mk.Matcher(patterns)
    om = ob.Matcher(patterns, True, 0, False)
    keep, rows, c, found = m.tag_records(recs, logging=True, filter_matching=True)
    keep_o, rows_o, c_o, found_o = ob.tag_records(om, recs, logging=True, filter_matching=True)
    assert (keep, rows, c) == (keep_o, rows_o, c_o)
    assert found == [sorted(set(f)) for f in found_o]
    for f, fo in list(zip(found, found_o))[:2000]:
        assert m.tag_value(f) == ob.tag_value(patterns, fo)
    assert sum(keep) >= 1000


@pytest.mark.parametrize("stride", [None, 4])
def test_config5_large_pattern_set(mk, stride):
    """500 k 21-mers: the set no longer fits the LDS filter; the filter moves to global memory
    (stride 8, 14-base q-grams, 4 MiB: the largest L2-resident geometry; stride 4 is the
    18-base alternative).  Correctness must hold regardless"""
    patterns = mk.parse_pattern_list(kmer_seq=_kmers(500_000, 21, 5))
    recs = _reads(20_000, 250, 51, plant=patterns, every=10)
    m = mk.Matcher(patterns, options=dict(force_stride=stride) if stride else None)
    assert m.use_ac and m.filter_mode()["in_lds"] is False
    if stride is None:
        assert m.filter_info()["stride"] == 8 and m.filter_info()["q_gram"] == 14
        assert m.filter_mode()["filter_bytes"] == 3 << 20
    else:
        assert m.filter_info()["stride"] == stride
    om = ob.Matcher(patterns, True, 0, False)
    assert m.extract_single(recs, logging=True) == ob.extract_single(om, recs, logging=True)


@pytest.mark.parametrize("stride", [1, 2, 4, 8, 16])
def test_global_filter_variants(mk, stride):
    """every kernel variant of the global-memory filter mode (forced on a small set)"""
    for k, n in ((31, 3000), (21, 2000), (16 + stride - 1, 500)):
        if stride > k:
            continue
        patterns = mk.parse_pattern_list(kmer_seq=_kmers(n, k, 60 + stride))
        recs = _reads(4000, 150, 70 + stride, plant=patterns, every=7)
        m = mk.Matcher(patterns, options=dict(force_stride=stride, force_global_filter=True))
        assert m.filter_mode()["in_lds"] is False and m.filter_info()["stride"] == stride
        om = ob.Matcher(patterns, True, 0, False)
        assert m.extract_single(recs, logging=True) == ob.extract_single(om, recs, logging=True)


def test_headline_full_size_properties(mk):
    """100 M x 150 bp, 10 k 31-mers, device-resident: any-mode == hits-mode, idempotence,
    shard additivity (two halves == whole), device counters consistent, and the oracle on
    slices of the identical host-generated bytes."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    lib = mk.load()
    patterns = mk.parse_pattern_list(kmer_seq=_kmers(10_000, 31, 6))
    m = mk.Matcher(patterns)
    n_rec, L, seed, every = 100_000_000, 150, 0xC0FFEE, 100
    n_bytes = n_rec * L
    st = torch.cuda.current_stream().cuda_stream
    d_seq = torch.empty(n_bytes + 64, dtype=torch.uint8, device=dev)
    d_off = torch.empty(n_rec + 1, dtype=torch.int64, device=dev)
    assert lib.mk_synth_reads_device(m.handle, seed, n_rec, L, every, d_seq.data_ptr(), d_off.data_ptr(), st) == 0
    npat = len(patterns)

    def scan(seq_t, off_t, n, mode, cap=0):
        flags = torch.empty((n + 7) // 4 * 4, dtype=torch.uint8, device=dev)
        hits = torch.empty(max(cap, 1) * 2, dtype=torch.int64, device=dev)
        nh = torch.zeros(1, dtype=torch.int64, device=dev)
        cnt = torch.zeros(npat + mk.MK_NUM_SUMMARY, dtype=torch.int64, device=dev)
        rc = lib.mk_scan_device(m.handle, seq_t.data_ptr(), n * L, off_t.data_ptr(), n, mode, flags.data_ptr(),
                                hits.data_ptr(), cap, nh.data_ptr(), cnt.data_ptr(), st)
        assert rc == 0, lib.mk_last_error()
        torch.cuda.synchronize()
        return flags[:n], hits, int(nh.item()), cnt.cpu().numpy()

    f_any, _, _, c_any = scan(d_seq, d_off, n_rec, mk.MK_MODE_ANY)
    cap = 4_000_000
    f_hit, d_hits, nh, c_hit = scan(d_seq, d_off, n_rec, mk.MK_MODE_HITS, cap)
    assert torch.equal(f_any, f_hit) and np.array_equal(c_any[npat:], c_hit[npat:])  # modes agree, run-to-run identical
    assert not c_any[:npat].any()  # occurrences per pattern exist in hits mode only (no-logging extract counts nothing)
    s = c_hit[npat:]
    n_flag = int(f_hit.sum(dtype=torch.int64).item())
    assert s[mk.MK_SUM_RECORDS] == n_rec and s[mk.MK_SUM_BASES] == n_bytes
    assert s[mk.MK_SUM_HITS] == nh == int(c_hit[:npat].sum()) and nh <= cap
    assert s[mk.MK_SUM_RECORDS_HIT] == n_flag
    assert n_rec // every * 0.9 < n_flag < n_rec // every * 1.1 + 1000
    hits = np.frombuffer(d_hits[:2 * nh].cpu().numpy().tobytes(), dtype=mk.HIT_DTYPE).copy()
    lib.mk_order_hits(m.handle, hits.ctypes.data, nh)
    assert np.all(np.diff(hits["rec"].astype(np.int64)) >= 0)  # sortedness
    assert len(np.unique(hits["rec"])) == n_flag               # every flagged record has a hit and vice versa
    assert np.all(hits["pos"] <= L - 31)

    # shard additivity: scanning [0, h) and [h, n) separately == the whole batch
    h = 50_000_008  # multiple of 8 records -> byte offset multiple of 16
    off_hi = (d_off[h:] - d_off[h]).contiguous()
    f_lo, _, _, c_lo = scan(d_seq, d_off, h, mk.MK_MODE_HITS, cap)
    f_hi, _, _, c_hi = scan(d_seq[h * L:], off_hi, n_rec - h, mk.MK_MODE_HITS, cap)
    assert torch.equal(torch.cat([f_lo, f_hi]), f_any)
    assert np.array_equal((c_lo + c_hi)[:npat], c_hit[:npat])
    for k in (mk.MK_SUM_HITS, mk.MK_SUM_RECORDS_HIT, mk.MK_SUM_RECORDS, mk.MK_SUM_BASES):
        assert c_lo[npat + k] + c_hi[npat + k] == c_any[npat + k]

    # oracle on slices of the identical bytes (host twin of the generator): start, middle, end
    om = ob.Matcher(patterns, True, 0, False)
    flags_np = None
    for rec0 in (0, 49_999_000, n_rec - 300_000):
        M = 300_000
        seq = np.zeros(M * L, dtype=np.uint8)
        off = np.zeros(M + 1, dtype=np.uint64)
        assert lib.mk_synth_reads_host(m.handle, seed, rec0, M, L, every, seq.ctypes.data, off.ctypes.data) == 0
        assert np.array_equal(d_seq[rec0 * L:(rec0 + M) * L].cpu().numpy(), seq)
        keep, c = ob.extract_single_packed(om, seq, off, logging=True, invert=False)
        flags_np = f_any[rec0:rec0 + M].cpu().numpy()
        assert np.array_equal(flags_np != 0, keep != 0)
        sel = (hits["rec"] >= rec0) & (hits["rec"] < rec0 + M)
        assert int(sel.sum()) == c["hits"][0]

    # THE LAUNCH bench.py TIMES: the same batch as a fixed-length batch -- mk_matcher_set_fixed_record_length(150) and
    # NO offsets array (d_seq_off = NULL).  Same flags as the offsets path (which the oracle has just checked on
    # three 300 k-record slices), same counters, and the same tuples in emission order.
    assert lib.mk_matcher_set_fixed_record_length(m.handle, L) == 0

    def scan_fixed(mode, cap=0):
        flags = torch.empty((n_rec + 7) // 4 * 4, dtype=torch.uint8, device=dev)
        hits_t = torch.empty(max(cap, 1) * 2, dtype=torch.int64, device=dev)
        nh_t = torch.zeros(1, dtype=torch.int64, device=dev)
        cnt = torch.zeros(npat + mk.MK_NUM_SUMMARY, dtype=torch.int64, device=dev)
        rc = lib.mk_scan_device(m.handle, d_seq.data_ptr(), n_bytes, None, n_rec, mode, flags.data_ptr(), hits_t.data_ptr(), cap,
                                nh_t.data_ptr(), cnt.data_ptr(), st)
        assert rc == 0, lib.mk_last_error()
        torch.cuda.synchronize()
        return flags[:n_rec], hits_t, int(nh_t.item()), cnt.cpu().numpy()

    del d_off
    fx_any, _, _, cx_any = scan_fixed(mk.MK_MODE_ANY)
    assert m.kernel_name == "mk_scan_kernel<8,24,false,false>"  # bench.py's headline kernel
    assert torch.equal(fx_any, f_any) and np.array_equal(cx_any, c_any)
    fx_hit, dx_hits, nhx, cx_hit = scan_fixed(mk.MK_MODE_HITS, cap)
    assert torch.equal(fx_hit, f_any) and nhx == nh and np.array_equal(cx_hit, c_hit)
    assert lib.mk_order_hits_device(m.handle, dx_hits.data_ptr(), nhx, st) == 0, lib.mk_last_error()
    torch.cuda.synchronize()
    hx = np.frombuffer(dx_hits[:2 * nhx].cpu().numpy().tobytes(), dtype=mk.HIT_DTYPE)
    assert np.array_equal(hx, hits)  # `hits`: the offsets path's tuples in the host's emission order
    with pytest.raises(AssertionError):  # a byte count that is not n_rec x 150 is refused, not scanned
        assert lib.mk_scan_device(m.handle, d_seq.data_ptr(), n_bytes - 1, None, n_rec, mk.MK_MODE_ANY, fx_any.data_ptr(), None, 0,
                                  torch.zeros(1, dtype=torch.int64, device=dev).data_ptr(), None, st) == 0
    assert lib.mk_matcher_set_fixed_record_length(m.handle, 0) == 0


def test_two_length_classes_full_size(mk):
    """the headline batch with ONE 8-mer added to the 10 000 31-mers (the set that used to drop to S = 1, q = 8): the
    two-class kernel flags the same records and counts the same occurrences as the one-class kernel at full size
    (15 GB), and the oracle agrees on a 300 k-record slice"""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    lib = mk.load()
    patterns = mk.parse_pattern_list(kmer_seq=_kmers(10_000, 31, 6) + [b"GATTACAG"])
    npat = len(patterns)
    n_rec, L, seed, every = 100_000_000, 150, 0xC1A55, 100
    st = torch.cuda.current_stream().cuda_stream
    d_seq = torch.empty(n_rec * L + 64, dtype=torch.uint8, device=dev)
    d_off = torch.empty(n_rec + 1, dtype=torch.int64, device=dev)
    res = {}
    for classes in (0, 1):
        m = mk.Matcher(patterns, options=dict(length_classes=classes))
        if classes == 0:
            assert lib.mk_synth_reads_device(m.handle, seed, n_rec, L, every, d_seq.data_ptr(), d_off.data_ptr(), st) == 0
            assert m.class_info() == {"split_len": 31, "n_short": 1, "q_gram2": 5, "stride2": 4}
        flags = torch.empty((n_rec + 7) // 4 * 4, dtype=torch.uint8, device=dev)
        nh = torch.zeros(1, dtype=torch.int64, device=dev)
        cnt = torch.zeros(npat + mk.MK_NUM_SUMMARY, dtype=torch.int64, device=dev)
        cap = 8_000_000
        d_hits = torch.empty(2 * cap, dtype=torch.int64, device=dev)
        assert lib.mk_scan_device(m.handle, d_seq.data_ptr(), n_rec * L, d_off.data_ptr(), n_rec, mk.MK_MODE_HITS, flags.data_ptr(),
                                  d_hits.data_ptr(), cap, nh.data_ptr(), cnt.data_ptr(), st) == 0, lib.mk_last_error()
        torch.cuda.synchronize()
        n = int(nh.item())
        assert n <= cap
        assert lib.mk_order_hits_device(m.handle, d_hits.data_ptr(), n, st) == 0, lib.mk_last_error()
        torch.cuda.synchronize()
        res[classes] = (m.kernel_name, flags[:n_rec].clone(), d_hits[:2 * n].clone(), cnt.cpu().numpy())
        if classes == 0:
            om = ob.Matcher(patterns, True, 0, False)
            M, rec0 = 300_000, 61_234_560
            seq = np.zeros(M * L, dtype=np.uint8)
            off = np.zeros(M + 1, dtype=np.uint64)
            assert lib.mk_synth_reads_host(m.handle, seed, rec0, M, L, every, seq.ctypes.data, off.ctypes.data) == 0
            keep, c = ob.extract_single_packed(om, seq, off, logging=True, invert=False)
            assert np.array_equal(flags[rec0:rec0 + M].cpu().numpy() != 0, keep != 0)
            assert int(keep.sum()) > M // every  # the 8-mer hits by chance as well (150 / 65536 per read)
        del m
    assert res[0][0].endswith("2-class>") and not res[1][0].endswith("2-class>")
    assert torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])
    skip = npat + mk.MK_SUM_CANDIDATES  # (the filter candidates are the one counter that depends on the geometry)
    assert np.array_equal(np.delete(res[0][3], skip), np.delete(res[1][3], skip))


def test_every_read_hits_full_size(mk):
    """100 M x 150 bp with a k-mer planted in EVERY read (tag on already extracted reads, at the headline
    size): the kernel variant for hit-dense text and the one for sparse hits flag the same records, the
    tuple count equals the counters, and 10^8 tuples sorted on the device (mk_order_hits_device) come out in
    emission order -- records ascending, positions ascending inside a record, every record present."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    lib = mk.load()
    patterns = mk.parse_pattern_list(kmer_seq=_kmers(10_000, 31, 6))
    npat = len(patterns)
    m = mk.Matcher(patterns)
    n_rec, L, seed = 100_000_000, 150, 0xD0D0
    st = torch.cuda.current_stream().cuda_stream
    d_seq = torch.empty(n_rec * L + 64, dtype=torch.uint8, device=dev)
    d_off = torch.empty(n_rec + 1, dtype=torch.int64, device=dev)
    assert lib.mk_synth_reads_device(m.handle, seed, n_rec, L, 1, d_seq.data_ptr(), d_off.data_ptr(), st) == 0
    cap = n_rec + (1 << 20)
    d_hits = torch.empty(2 * cap, dtype=torch.int64, device=dev)
    d_nh = torch.zeros(1, dtype=torch.int64, device=dev)
    flags, names = {}, {}
    for density in (1000, 0):
        f = torch.empty(n_rec + 8, dtype=torch.uint8, device=dev)
        cnt = torch.zeros(npat + mk.MK_NUM_SUMMARY, dtype=torch.int64, device=dev)
        assert lib.mk_matcher_hint_hit_density(m.handle, density) == 0
        mode = mk.MK_MODE_HITS if density else mk.MK_MODE_ANY
        assert lib.mk_scan_device(m.handle, d_seq.data_ptr(), n_rec * L, d_off.data_ptr(), n_rec, mode, f.data_ptr(),
                                  d_hits.data_ptr(), cap if density else 0, d_nh.data_ptr(), cnt.data_ptr(), st) == 0, lib.mk_last_error()
        torch.cuda.synchronize()
        flags[density], names[density] = f[:n_rec], m.kernel_name
        if density:
            nh = int(d_nh.item())
            c = cnt.cpu().numpy()
    assert names[1000].endswith("plain>") and not names[0].endswith("plain>")
    assert torch.equal(flags[1000], flags[0]) and bool(flags[0].all())
    assert n_rec <= nh <= cap and c[npat + mk.MK_SUM_HITS] == nh == int(c[:npat].sum()) and c[npat + mk.MK_SUM_RECORDS_HIT] == n_rec
    assert lib.mk_order_hits_device(m.handle, d_hits.data_ptr(), nh, st) == 0, lib.mk_last_error()
    torch.cuda.synchronize()
    t = d_hits[:2 * nh].view(nh, 2)  # [rec, pat | pos << 32]
    rec, pos = t[:, 0], t[:, 1] >> 32
    d_rec = rec[1:] - rec[:-1]
    assert bool((d_rec >= 0).all())
    assert bool(((d_rec > 0) | (pos[1:] >= pos[:-1])).all())  # equal-length patterns: end order == start order
    assert int((d_rec > 0).sum().item()) + 1 == n_rec and int(rec[0].item()) == 0 and int(rec[-1].item()) == n_rec - 1
    assert bool((pos <= L - 31).all()) and bool(((t[:, 1] & 0xFFFFFFFF) < npat).all())


@pytest.mark.parametrize("n_pat,every", [(1, 1000), (1, 50), (13, 100000)])
def test_sparse_candidates_full_size(mk, n_pat, every):
    """few patterns on a 15 GB batch: filter positives are so rare that a wave's ring never fills
    by itself, and a queued position (32 bits) must not outlive 4 GiB of the wave's advance.
    Shards below 4 GiB cannot have the problem: the whole batch must equal their concatenation.
    (Regression: 1 pattern, 1 read in 1000 planted lost two hits in three.)"""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    lib = mk.load()
    patterns = mk.parse_pattern_list(kmer_seq=_kmers(n_pat, 31, 16))
    m = mk.Matcher(patterns)
    n_rec, L, seed = 100_000_000, 150, 0xBEEF + n_pat
    st = torch.cuda.current_stream().cuda_stream
    d_seq = torch.empty(n_rec * L + 64, dtype=torch.uint8, device=dev)
    d_off = torch.empty(n_rec + 1, dtype=torch.int64, device=dev)
    assert lib.mk_synth_reads_device(m.handle, seed, n_rec, L, every, d_seq.data_ptr(), d_off.data_ptr(), st) == 0
    npat = len(patterns)

    def scan(seq_t, off_t, n):
        flags = torch.empty((n + 7) // 4 * 4, dtype=torch.uint8, device=dev)
        nh = torch.zeros(1, dtype=torch.int64, device=dev)
        cnt = torch.zeros(npat + mk.MK_NUM_SUMMARY, dtype=torch.int64, device=dev)
        rc = lib.mk_scan_device(m.handle, seq_t.data_ptr(), n * L, off_t.data_ptr(), n, mk.MK_MODE_ANY, flags.data_ptr(),
                                None, 0, nh.data_ptr(), cnt.data_ptr(), st)
        assert rc == 0, lib.mk_last_error()
        torch.cuda.synchronize()
        return flags[:n], int(nh.item()), cnt.cpu().numpy()

    f_all, nh_all, c_all = scan(d_seq, d_off, n_rec)
    shard = 12_500_000  # 1.875 GB of text, a multiple of 8 records (16-byte aligned start)
    parts, nh_sum, c_sum = [], 0, np.zeros_like(c_all)
    for b in range(0, n_rec, shard):
        off_s = (d_off[b:b + shard + 1] - d_off[b]).contiguous()
        f, nh, c = scan(d_seq[b * L:], off_s, shard)
        parts.append(f)
        nh_sum += nh
        c_sum += c
    assert nh_all == nh_sum == 0  # *d_n_hits is the tuple count of MK_MODE_HITS
    assert np.array_equal(c_all[:npat], c_sum[:npat])
    for k in (mk.MK_SUM_HITS, mk.MK_SUM_RECORDS_HIT, mk.MK_SUM_RECORDS, mk.MK_SUM_BASES):
        assert c_all[npat + k] == c_sum[npat + k]
    assert not c_all[:npat].any()  # per-pattern occurrences are a MK_MODE_HITS output
    assert torch.equal(torch.cat(parts), f_all)
    n_flag = int(f_all.sum(dtype=torch.int64).item())
    assert n_rec // every * 0.9 - 10 < n_flag < n_rec // every * 1.1 + 10


def test_ragged_records_full_size(mk):
    """14.4 GB of text cut into 100 M records of eight different lengths: the record lookup of a
    verified hit starts from an interpolated index that is off by a few records here, so the
    gallop / bisect path runs at full scale.  Whole batch == concatenation of 1.8 GB shards, and the
    oracle agrees on the first records."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    lib = mk.load()
    patterns = mk.parse_pattern_list(kmer_seq=_kmers(2000, 31, 26))
    m = mk.Matcher(patterns)
    npat = len(patterns)
    cycle = [100, 151, 200, 149, 120, 180, 150, 102]  # 1152 bytes = 72 x 16
    n_cyc = 12_500_000
    n_rec, n_bytes = 8 * n_cyc, 1152 * n_cyc
    st = torch.cuda.current_stream().cuda_stream
    d_seq = torch.empty(n_bytes + 64, dtype=torch.uint8, device=dev)
    d_off_u = torch.empty(n_bytes // 150 + 1, dtype=torch.int64, device=dev)
    # uniform 150-byte synthetic reads with planted k-mers, then re-cut at the ragged offsets
    assert lib.mk_synth_reads_device(m.handle, 0xFACE, n_bytes // 150, 150, 20, d_seq.data_ptr(), d_off_u.data_ptr(), st) == 0
    del d_off_u
    lens = torch.tensor(cycle, dtype=torch.int64, device=dev).repeat(n_cyc)
    d_off = torch.zeros(n_rec + 1, dtype=torch.int64, device=dev)
    torch.cumsum(lens, 0, out=d_off[1:])
    del lens
    assert int(d_off[-1].item()) == n_bytes

    def scan(seq_t, off_t, n, nb):
        flags = torch.empty((n + 7) // 4 * 4, dtype=torch.uint8, device=dev)
        nh = torch.zeros(1, dtype=torch.int64, device=dev)
        cnt = torch.zeros(npat + mk.MK_NUM_SUMMARY, dtype=torch.int64, device=dev)
        rc = lib.mk_scan_device(m.handle, seq_t.data_ptr(), nb, off_t.data_ptr(), n, mk.MK_MODE_ANY, flags.data_ptr(),
                                None, 0, nh.data_ptr(), cnt.data_ptr(), st)
        assert rc == 0, lib.mk_last_error()
        torch.cuda.synchronize()
        return flags[:n], cnt.cpu().numpy()

    f_all, c_all = scan(d_seq, d_off, n_rec, n_bytes)
    parts, c_sum = [], np.zeros_like(c_all)
    shard_c = n_cyc // 8  # cycles per shard
    for k in range(8):
        r0, b0 = k * shard_c * 8, k * shard_c * 1152
        off_s = (d_off[r0:r0 + shard_c * 8 + 1] - b0).contiguous()
        f, c = scan(d_seq[b0:], off_s, shard_c * 8, shard_c * 1152)
        parts.append(f)
        c_sum += c
    assert torch.equal(torch.cat(parts), f_all)
    assert np.array_equal(c_all[:npat], c_sum[:npat])
    for k in (mk.MK_SUM_HITS, mk.MK_SUM_RECORDS_HIT, mk.MK_SUM_RECORDS, mk.MK_SUM_BASES):
        assert c_all[npat + k] == c_sum[npat + k]
    assert c_all[npat + mk.MK_SUM_RECORDS_HIT] == int(f_all.sum(dtype=torch.int64).item()) > n_rec // 40
    # oracle on the first 160 k records
    M = 160_000
    off = d_off[:M + 1].cpu().numpy().astype(np.uint64)
    seq = d_seq[:int(off[-1])].cpu().numpy()
    om = ob.Matcher(patterns, True, 0, False)
    keep, _ = ob.extract_single_packed(om, seq, off, logging=False, invert=False)
    assert np.array_equal(f_all[:M].cpu().numpy() != 0, keep != 0)


def test_config5_full_size_shard_additivity(mk):
    """BASELINE config 5 at a full per-GPU shard: 500 k 21-mers (global-memory filter) over
    50 M x 250 bp = 12.5 GB.  Whole batch == concatenation of four 3.1 GB shards; hits mode agrees
    with flags mode."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    lib = mk.load()
    patterns = mk.parse_pattern_list(kmer_seq=_kmers(500_000, 21, 5))
    m = mk.Matcher(patterns)
    assert m.filter_mode()["in_lds"] is False
    npat = len(patterns)
    n_rec, L = 50_000_000, 250
    st = torch.cuda.current_stream().cuda_stream
    d_seq = torch.empty(n_rec * L + 64, dtype=torch.uint8, device=dev)
    d_off = torch.empty(n_rec + 1, dtype=torch.int64, device=dev)
    assert lib.mk_synth_reads_device(m.handle, 0x5EED, n_rec, L, 100, d_seq.data_ptr(), d_off.data_ptr(), st) == 0

    def scan(seq_t, off_t, n, mode=None, cap=0):
        mode = mk.MK_MODE_ANY if mode is None else mode
        flags = torch.empty((n + 7) // 4 * 4, dtype=torch.uint8, device=dev)
        hits = torch.empty(max(cap, 1) * 2, dtype=torch.int64, device=dev)
        nh = torch.zeros(1, dtype=torch.int64, device=dev)
        cnt = torch.zeros(npat + mk.MK_NUM_SUMMARY, dtype=torch.int64, device=dev)
        rc = lib.mk_scan_device(m.handle, seq_t.data_ptr(), n * L, off_t.data_ptr(), n, mode, flags.data_ptr(),
                                hits.data_ptr(), cap, nh.data_ptr(), cnt.data_ptr(), st)
        assert rc == 0, lib.mk_last_error()
        torch.cuda.synchronize()
        return flags[:n], int(nh.item()), cnt.cpu().numpy()

    f_all, _, c_all = scan(d_seq, d_off, n_rec)
    f_hit, nh, c_hit = scan(d_seq, d_off, n_rec, mk.MK_MODE_HITS, 2_000_000)
    assert torch.equal(f_all, f_hit) and np.array_equal(c_all[npat:], c_hit[npat:]) and nh == c_hit[npat + mk.MK_SUM_HITS]
    assert nh == int(c_hit[:npat].sum())
    shard = 12_500_000  # x 250 B: a multiple of 16 bytes
    parts, c_sum = [], np.zeros_like(c_all)
    for b in range(0, n_rec, shard):
        off_s = (d_off[b:b + shard + 1] - d_off[b]).contiguous()
        f, _, c = scan(d_seq[b * L:], off_s, shard, mk.MK_MODE_HITS, 2_000_000)
        parts.append(f)
        c_sum += c
    assert torch.equal(torch.cat(parts), f_all)
    assert np.array_equal(c_hit[:npat], c_sum[:npat])
    for k in (mk.MK_SUM_HITS, mk.MK_SUM_RECORDS_HIT, mk.MK_SUM_RECORDS, mk.MK_SUM_BASES):
        assert c_all[npat + k] == c_sum[npat + k]
    n_flag = int(f_all.sum(dtype=torch.int64).item())
    assert n_rec // 100 * 0.9 < n_flag < n_rec // 100 * 1.2


def test_config4_full_size_bam_through_the_cli(mk, tmp_path):
    """BASELINE config 4 end to end: a synthetic 20 M-record BAM (150 bp, 4-bit sequences, BGZF) through
    `merkurio tag -m` with 10 k 31-mers.  The codec side (parallel BGZF inflate, un-nibbling, batched
    scan, tag append, BAM pass-through) must agree with the library called directly on the same
    sequences: the kept records, in order, and every km value.  MERKURIO_C4_RECORDS scales it down."""
    import gzip
    import os
    import struct
    import subprocess
    import zlib
    from concurrent.futures import ThreadPoolExecutor
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cli = os.path.join(root, "merkurio_amd", "lib", "merkurio")
    n, L, every = int(os.environ.get("MERKURIO_C4_RECORDS", "20000000")), 150, 100
    patterns = mk.parse_pattern_list(kmer_seq=_kmers(10_000, 31, 44))
    (tmp_path / "k.txt").write_bytes(b"\n".join(patterns) + b"\n")
    pat_arr = np.frombuffer(b"".join(patterns), dtype=np.uint8).reshape(len(patterns), 31)
    rec_bytes = 4 + 32 + 11 + (L + 1) // 2 + L
    fixed = struct.pack("<iiiBBHHHiiii", rec_bytes - 4, -1, -1, 11, 0, 4680, 0, 4, L, -1, -1, 0)
    nib = np.zeros(256, dtype=np.uint8)
    for ch, v in zip(b"ACGT", (1, 2, 4, 8)):
        nib[ch] = v
    eof = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")

    def bgzf_block(chunk):
        co = zlib.compressobj(1, zlib.DEFLATED, -15)
        c = co.compress(chunk) + co.flush()
        return (bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0]) + struct.pack("<H", len(c) + 25) + c +
                struct.pack("<II", zlib.crc32(chunk), len(chunk)))

    header_text = b"@HD\tVN:1.6\tSO:unsorted\n"
    head = b"BAM\x01" + struct.pack("<i", len(header_text)) + header_text + struct.pack("<i", 0)
    seq_all = np.zeros(n * L + 1, dtype=np.uint8)  # the ASCII sequences the matcher must see (+1 pad byte)
    slab = 1_000_000
    rng = np.random.default_rng(2024)
    with open(tmp_path / "big.bam", "wb") as f, ThreadPoolExecutor(16) as ex:
        f.write(bgzf_block(head))
        for s0 in range(0, n, slab):
            m_ = min(slab, n - s0)
            seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(m_, L), dtype=np.uint8)]
            idx = np.arange(s0, s0 + m_)
            planted = idx[idx % every == 7] - s0
            which = (idx[planted] * 2654435761) % len(patterns)
            offs = (idx[planted] * 40503) % (L - 31 + 1)
            for r, w, o in zip(planted.tolist(), which.tolist(), offs.tolist()):
                seq[r, o:o + 31] = pat_arr[w]
            seq_all[s0 * L:(s0 + m_) * L] = seq.reshape(-1)
            rec = np.empty((m_, rec_bytes), dtype=np.uint8)
            rec[:, :36] = np.frombuffer(fixed, dtype=np.uint8)
            rec[:, 36] = ord("r")
            digits = idx.copy()
            for d in range(9, 0, -1):
                rec[:, 36 + d] = 48 + digits % 10
                digits //= 10
            rec[:, 46] = 0
            nb = nib[seq]
            rec[:, 47:47 + L // 2] = (nb[:, 0::2] << 4) | nb[:, 1::2]
            rec[:, 47 + L // 2:] = 40
            raw = rec.tobytes()
            per = (0xff00 // rec_bytes) * rec_bytes  # whole records per BGZF block
            for blk in ex.map(bgzf_block, [raw[b:b + per] for b in range(0, len(raw), per)]):
                f.write(blk)
        f.write(eof)
    p = subprocess.run([cli, "tag", "-i", str(tmp_path / "big.bam"), "-f", str(tmp_path / "k.txt"), "-m", "-o",
                        str(tmp_path / "out.bam")], capture_output=True)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    # expected: the library on the same sequences (device path), whole batch at once
    m = mk.Matcher(patterns)
    off = np.arange(n + 1, dtype=np.uint64) * L
    flags, hits = m.scan_packed(seq_all, off, mk.MK_MODE_HITS, hits_cap=max(1 << 20, 4 * n // every))
    want = {}
    for r, pt in zip(hits["rec"].tolist(), hits["pat"].tolist()):
        want.setdefault(r, set()).add(pt)
    assert sorted(want) == np.flatnonzero(flags).tolist() and len(want) >= n // every
    # got: records of out.bam in order (name, km value)
    raw = gzip.decompress(open(tmp_path / "out.bam", "rb").read())
    assert raw[:4] == b"BAM\x01"
    l_text = struct.unpack_from("<i", raw, 4)[0]
    pos = 8 + l_text + 4
    got = []
    while pos < len(raw):
        bs = struct.unpack_from("<i", raw, pos)[0]
        body = raw[pos + 4:pos + 4 + bs]
        name = body[32:32 + 10]
        aux = body[32 + 11 + (L + 1) // 2 + L:]
        assert aux[:3] == b"kmZ" and aux[-1:] == b"\0"
        got.append((int(name[1:]), aux[3:-1]))
        pos += 4 + bs
    assert [g[0] for g in got] == sorted(want)
    for r, val in got[:: max(1, len(got) // 5000)]:
        assert val == b",".join(patterns[k] for k in sorted(want[r])), r
    # ... and against the ORACLE (the reference's tag loop restated, src/cmd_tag.rs:387-490) on a 200 k-record
    # slice of the same BAM: which records `-m` keeps and the km value of each of them
    a = min(n, 7_300_000) - min(n, 200_000) if n >= 400_000 else 0
    b = min(n, a + 200_000)
    recs = [seq_all[i * L:(i + 1) * L].tobytes() for i in range(a, b)]
    om = ob.Matcher(patterns, True, 0, False)
    keep_o, _, _, found_o = ob.tag_records(om, recs, logging=False, filter_matching=True)
    by_rec = dict(got)
    kept_cli = [r for r, _ in got if a <= r < b]
    assert kept_cli == [a + i for i, k in enumerate(keep_o) if k] and len(kept_cli) >= (b - a) // every - 1
    for i, k in enumerate(keep_o):
        if k:
            assert by_rec[a + i] == ob.tag_value(patterns, found_o[i]), a + i


def test_more_than_2_pow_32_records(mk):
    """2^32 + 16 records (one byte each: 4.3 GB of text, 34 GB of offsets): record indices beyond 32 bits through
    every place that narrows them elsewhere -- no per-wave flag lists (their entries are 32 bits), no coarse record
    index for ragged batches (its entries are 32 bits), 64-bit records in the tuples and in the emission order.
    The expectation is by construction: the text is all 'C' but for 'A' in chosen records."""
    torch = pytest.importorskip("torch")
    lib = mk.load()
    dev = torch.device("cuda", 0)
    n = (1 << 32) + 16
    planted = [0, 5, (1 << 31) - 1, 1 << 31, (1 << 32) - 1, 1 << 32, (1 << 32) + 7, n - 1]
    m = mk.Matcher([b"A"])  # one pattern: the BNDMq domain, stride 1
    st = torch.cuda.current_stream().cuda_stream
    d_flags = torch.empty(n + 8, dtype=torch.uint8, device=dev)  # (every scan clears it)
    d_hits = torch.zeros(2 * 1024, dtype=torch.int64, device=dev)
    d_nh = torch.zeros(1, dtype=torch.int64, device=dev)
    d_cnt = torch.zeros(1 + mk.MK_NUM_SUMMARY, dtype=torch.int64, device=dev)

    def check(d_seq, n_bytes, d_off, where, label):
        d_cnt.zero_()
        for mode in (mk.MK_MODE_ANY, mk.MK_MODE_HITS):
            assert lib.mk_scan_device(m.handle, d_seq.data_ptr(), n_bytes, d_off.data_ptr() if d_off is not None else None, n, mode,
                                      d_flags.data_ptr(), d_hits.data_ptr(), 1024, d_nh.data_ptr(), d_cnt.data_ptr(), st) == 0, lib.mk_last_error()
            torch.cuda.synchronize()
            flagged = [r for r in where if int(d_flags[r:r + 1].item()) == 1]
            assert flagged == where, (label, mode, flagged)
            # (sums in pieces of 2^30: nothing here relies on 64-bit indexing inside a torch reduction)
            total = sum(int(d_flags[a:min(n, a + (1 << 30))].sum(dtype=torch.int64).item()) for a in range(0, n, 1 << 30))
            assert total == len(where), (label, mode, total)
        assert lib.mk_matcher_check_device(m.handle, st) == 0
        nh = int(d_nh.item())
        assert nh == len(where)
        assert lib.mk_order_hits_device(m.handle, d_hits.data_ptr(), nh, st) == 0, lib.mk_last_error()
        torch.cuda.synchronize()
        got = np.frombuffer(d_hits.cpu().numpy().tobytes(), dtype=mk.HIT_DTYPE)[:nh]
        assert got["rec"].tolist() == sorted(where) and set(got["pat"].tolist()) == {0}, label
        c = d_cnt.cpu().numpy()
        assert c[0] == len(where) and c[1 + mk.MK_SUM_RECORDS_HIT] == 2 * len(where) and c[1 + mk.MK_SUM_RECORDS] == 2 * n
        return got

    # (1) equal lengths through the offsets array
    # torch's own kernels index with 32 bits in places (arange over 2^32 + 17 elements returned zeros beyond 2^32 on
    # this image): every torch operation below works on pieces of at most 2^30 elements
    PIECE = 1 << 30

    def fill(t, value):
        for a in range(0, t.numel(), PIECE):
            t[a:a + PIECE].fill_(value)

    def iota(t, add_from=None):  # t[i] = i (+ 1 from index add_from on)
        for a in range(0, t.numel(), PIECE):
            b = min(t.numel(), a + PIECE)
            t[a:b] = torch.arange(a, b, dtype=torch.int64, device=dev)
            if add_from is not None and b > add_from:
                t[max(a, add_from):b] += 1

    def plant(t, positions):  # one slice per byte
        for q in positions:
            t[q:q + 1] = ord("A")
        torch.cuda.synchronize()
        assert all(int(t[q:q + 1].item()) == ord("A") for q in positions)

    d_seq = torch.empty(n + 64, dtype=torch.uint8, device=dev)
    fill(d_seq, ord("C"))
    plant(d_seq, planted)
    assert int(d_seq[(1 << 32) + 1:(1 << 32) + 2].item()) == ord("C")
    d_off = torch.empty(n + 1, dtype=torch.int64, device=dev)
    iota(d_off)
    assert int(d_off[n:n + 1].item()) == n and int(d_off[(1 << 32) + 3:(1 << 32) + 4].item()) == (1 << 32) + 3
    got = check(d_seq, n, d_off, planted, "offsets")
    assert set(got["pos"].tolist()) == {0}
    # (2) the same as a fixed-length batch: no offsets array
    assert lib.mk_matcher_set_fixed_record_length(m.handle, 1) == 0
    check(d_seq, n, None, planted, "fixed length")
    assert lib.mk_matcher_set_fixed_record_length(m.handle, 0) == 0
    # (3) ragged: record 3 has two bytes ("CA": an occurrence at position 1), every record behind it starts one byte later
    iota(d_off, add_from=4)
    d_seq2 = torch.empty(n + 1 + 64, dtype=torch.uint8, device=dev)
    fill(d_seq2, ord("C"))
    where = [3] + [r for r in planted if r > 3]
    plant(d_seq2, [4] + [r + 1 for r in planted if r > 3])
    assert int(d_off[n:n + 1].item()) == n + 1 and int(d_off[3:4].item()) == 3 and int(d_off[4:5].item()) == 5
    assert lib.mk_matcher_hint_record_lengths(m.handle, 0) == 0
    got = check(d_seq2, n + 1, d_off, where, "ragged")
    assert got["pos"].tolist() == [1] + [0] * (len(where) - 1)


def test_occurrence_beyond_4_gib_of_its_record_is_reported(mk):
    """mk_hit.pos is 32 bits.  mk_scan_batch refuses a record of 4 GiB or more before it scans; the device API only
    enqueues, so its tuple kernels raise a sticky error word and mk_matcher_check_device / mk_order_hits_device
    return MK_E_UNSUPPORTED -- never a wrapped position.  The flags of that scan are still right."""
    torch = pytest.importorskip("torch")
    lib = mk.load()
    dev = torch.device("cuda", 0)
    patterns = mk.parse_pattern_list(kmer_seq=_kmers(20, 31, 9))
    m = mk.Matcher(patterns)
    big = (1 << 32) + 4096
    lens = [1000, big, 500]
    n_bytes = sum(lens)
    d_seq = torch.full((n_bytes + 64,), ord("C"), dtype=torch.uint8, device=dev)
    p0 = torch.frombuffer(bytearray(patterns[0]), dtype=torch.uint8).to(dev)
    d_off = torch.tensor([0, 1000, 1000 + big, n_bytes], dtype=torch.int64, device=dev)
    d_flags = torch.zeros(8, dtype=torch.uint8, device=dev)
    d_hits = torch.zeros(2 * 64, dtype=torch.int64, device=dev)
    d_nh = torch.zeros(1, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def scan():
        assert lib.mk_scan_device(m.handle, d_seq.data_ptr(), n_bytes, d_off.data_ptr(), 3, mk.MK_MODE_HITS, d_flags.data_ptr(),
                                  d_hits.data_ptr(), 64, d_nh.data_ptr(), None, st) == 0, lib.mk_last_error()

    # occurrences in the first 4 GiB of the long record and in the short ones: fine
    for o in (10, 1000 + 77, 1000 + (1 << 32) - 31, 1000 + big + 400):
        d_seq[o:o + 31] = p0
    scan()
    assert lib.mk_matcher_check_device(m.handle, st) == 0
    assert int(d_nh.item()) == 4 and d_flags[:3].tolist() == [1, 1, 1]
    assert lib.mk_order_hits_device(m.handle, d_hits.data_ptr(), 4, st) == 0
    torch.cuda.synchronize()
    got = np.frombuffer(d_hits.cpu().numpy().tobytes(), dtype=mk.HIT_DTYPE)[:4]
    assert list(zip(got["rec"].tolist(), got["pos"].tolist())) == [(0, 10), (1, 77), (1, (1 << 32) - 31), (2, 400)]
    # one more, 4 GiB + 100 into the long record: reported, once
    d_seq[1000 + (1 << 32) + 100:1000 + (1 << 32) + 131] = p0
    scan()
    assert lib.mk_matcher_check_device(m.handle, st) == mk.MK_E_UNSUPPORTED
    assert b"4 GiB" in lib.mk_last_error()
    assert lib.mk_matcher_check_device(m.handle, st) == 0  # cleared by the call that reported it
    assert d_flags[:3].tolist() == [1, 1, 1]
    scan()
    assert lib.mk_order_hits_device(m.handle, d_hits.data_ptr(), int(d_nh.item()), st) == mk.MK_E_UNSUPPORTED
    # flags-only scans have no position to lose
    assert lib.mk_scan_device(m.handle, d_seq.data_ptr(), n_bytes, d_off.data_ptr(), 3, mk.MK_MODE_ANY, d_flags.data_ptr(),
                              None, 0, d_nh.data_ptr(), None, st) == 0
    assert lib.mk_matcher_check_device(m.handle, st) == 0 and d_flags[:3].tolist() == [1, 1, 1]

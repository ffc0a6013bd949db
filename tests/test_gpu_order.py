"""Emission order on the device (mk_order_hits_device, order_hits.hip) against the reference's comparator applied
with numpy on the host: Aho-Corasick order (record, end ascending, longer pattern first, pattern id --
src/cmd_extract.rs:332-351) and BNDMq order (record, pattern, start -- src/cmd_extract.rs:365-384), on tuple
sets that walk every path of the device code: bins of consecutive records, re-binning on (record, end) for few
huge records, the re-binning on the whole key, the library fallback, shuffled input, tuples of a batch the handle has not scanned, and every
small size around the leaf-sort geometry.  The scan-produced tuples of real batches are covered by
test_gpu_parity.py::test_emission_order_on_the_device and test_gpu_configs.py (10^8 tuples)."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mk():
    from merkurio_amd import native
    native.load()
    if native.device_count() < 1:
        pytest.fail("no HIP device visible: the gpu-marked tests need an MI355X")
    return native


def _patterns(rnd, n, lens):
    out = set()
    while len(out) < n:
        out.add(bytes(rnd.choice(b"ACGT") for _ in range(rnd.choice(lens))))
    return sorted(out)


def _expected(h, ac, plen):
    """the reference's emission order of the tuple set h"""
    rec, pat, pos = h["rec"].astype(np.uint64), h["pat"].astype(np.int64), h["pos"].astype(np.int64)
    if ac:
        order = np.lexsort((pat, pos, pos + plen[pat], rec))  # last key is the primary one
    else:
        order = np.lexsort((pos, pat, rec))
    return h[order]


def _order_on_device(mk, m, h):
    import torch
    lib = mk.load()
    dev = torch.device("cuda", 0)
    n = len(h)
    d = torch.from_numpy(np.frombuffer(h.tobytes(), dtype=np.int64).copy()).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    assert lib.mk_order_hits_device(m.handle, d.data_ptr(), n, st) == 0, lib.mk_last_error()
    torch.cuda.synchronize()
    return np.frombuffer(d.cpu().numpy().tobytes(), dtype=mk.HIT_DTYPE)[:n]


def _tuples(mk, rnd, n, n_rec, max_pos, n_pat, rec_of=None):
    """n distinct (rec, pat, pos) tuples"""
    g = np.random.default_rng(rnd.randrange(1 << 30))
    rec = g.integers(0, n_rec, size=2 * n, dtype=np.uint64) if rec_of is None else rec_of(g, 2 * n)
    pat = g.integers(0, n_pat, size=2 * n, dtype=np.uint32)
    pos = g.integers(0, max_pos, size=2 * n, dtype=np.uint32)
    h = np.zeros(2 * n, dtype=mk.HIT_DTYPE)
    h["rec"], h["pat"], h["pos"] = rec, pat, pos
    h = np.unique(h)  # distinct tuples: the scan never reports an occurrence twice
    g.shuffle(h)
    return h[:n].copy()


CASES = [
    # name, algo, pattern lengths, n tuples, n records, max pos, expected path (None = any)
    ("kmers-uniform", "ac", [31], 300_000, 1_000_000, 120, 1),
    ("kmers-dense", "ac", [21], 1_500_000, 400_000, 230, 1),
    ("mixed-lengths", "ac", [3, 4, 9, 21, 31, 64, 100], 250_000, 50_000, 400, 1),
    ("bndmq", "bndmq", [5, 9, 31, 64], 200_000, 300_000, 300, 1),
    ("few-huge-records-ac", "ac", [31], 600_000, 5, 1 << 27, 2),
    ("few-huge-records-mixed", "ac", [4, 31, 90], 400_000, 3, 1 << 26, 2),
    ("few-huge-records-bndmq", "bndmq", [7, 31], 300_000, 2, 1 << 27, None),
    ("one-record", "ac", [31], 100_000, 1, 3_000_000, 2),
]


@pytest.mark.parametrize("name,algo,lens,n,n_rec,max_pos,path", CASES, ids=[c[0] for c in CASES])
def test_order_matches_the_reference_comparator(mk, name, algo, lens, n, n_rec, max_pos, path):
    rnd = random.Random(hash(name) & 0xFFFF)
    n_pat = 13 if algo == "bndmq" else 700
    pats = _patterns(rnd, n_pat, lens)
    m = mk.Matcher(pats, algo=mk.MK_ALGO_AC if algo == "ac" else mk.MK_ALGO_BNDMQ)
    plen = np.array([len(p) for p in pats], dtype=np.int64)
    h = _tuples(mk, rnd, n, n_rec, max_pos, len(pats))
    got = _order_on_device(mk, m, h)
    info = m.order_info()
    assert np.array_equal(got, _expected(h, algo == "ac", plen)), info
    if path is not None:
        assert info["path"] == path, info
    assert info["max_bin"] <= 16384


def test_order_every_small_size(mk):
    """sizes around the leaf geometry (16 keys per lane, 64..1024 lanes) and the first bin split"""
    rnd = random.Random(5)
    pats = _patterns(rnd, 40, [6, 31, 33])
    m = mk.Matcher(pats, algo=mk.MK_ALGO_AC)
    plen = np.array([len(p) for p in pats], dtype=np.int64)
    for n in (2, 3, 15, 16, 17, 63, 64, 65, 1000, 1023, 1024, 1025, 2047, 2048, 2049, 4095, 4097, 9000, 16383, 16384, 16385, 40000):
        h = _tuples(mk, rnd, n, max(2, n // 3), 180, len(pats))
        assert len(h) == n
        got = _order_on_device(mk, m, h)
        assert np.array_equal(got, _expected(h, True, plen)), (n, m.order_info())
        assert m.order_info()["path"] in (1, 2)


def test_order_clustered_hits_and_foreign_batches(mk):
    """(a) every tuple inside 64 consecutive records of a batch of 2^24 (the record bins overflow -> bins on
    (record, end)); (b) tuples of a batch the handle has not scanned: records far beyond its last scan's count"""
    import torch
    rnd = random.Random(9)
    pats = _patterns(rnd, 300, [31])
    m = mk.Matcher(pats)
    plen = np.array([31] * len(pats), dtype=np.int64)
    lib = mk.load()
    dev = torch.device("cuda", 0)
    # a small scan fixes the handle's record bound at 1000 records
    recs = [bytes(rnd.choice(b"ACGT") for _ in range(100)) for _ in range(1000)]
    m.scan(recs, mk.MK_MODE_ANY)
    h = _tuples(mk, rnd, 200_000, 1 << 24, 200, len(pats))  # (b)
    got = _order_on_device(mk, m, h)
    assert np.array_equal(got, _expected(h, True, plen)), m.order_info()
    assert m.order_info()["path"] == 1
    h = _tuples(mk, rnd, 150_000, 0, 150, len(pats),
                rec_of=lambda g, k: (np.uint64((1 << 24) - 70) + g.integers(0, 64, size=k, dtype=np.uint64)))  # (a)
    got = _order_on_device(mk, m, h)
    assert np.array_equal(got, _expected(h, True, plen)), m.order_info()
    assert m.order_info()["path"] == 2


def test_order_bins_on_the_whole_key_when_record_and_end_do_not_split(mk):
    """tuples that share record AND end position (here 20 000 patterns ending on one byte), or one pattern all over one
    long record under BNDMq order (record and pattern constant): the re-binning takes the top bits of the whole
    (record, A, B) triple, so the last field splits them"""
    rnd = random.Random(3)
    pats = _patterns(rnd, 20_000, [12])
    m = mk.Matcher(pats)
    plen = np.array([12] * len(pats), dtype=np.int64)
    h = np.zeros(20_000 + 5_000, dtype=mk.HIT_DTYPE)
    h["rec"][:20_000], h["pat"][:20_000], h["pos"][:20_000] = 7, np.arange(20_000), 5
    h["rec"][20_000:], h["pat"][20_000:], h["pos"][20_000:] = np.arange(5_000) % 9, np.arange(5_000), 6 + np.arange(5_000) % 50
    np.random.default_rng(1).shuffle(h)
    got = _order_on_device(mk, m, h)
    assert np.array_equal(got, _expected(h, True, plen))
    assert m.order_info()["path"] == 2 and m.order_info()["max_bin"] <= 16384, m.order_info()
    # one pattern, one record, 100 000 positions, BNDMq order
    mb = mk.Matcher([b"AAC"], algo=mk.MK_ALGO_BNDMQ)
    pos = np.random.default_rng(2).choice(3_000_000, size=100_000, replace=False).astype(np.uint32)
    h = np.zeros(len(pos), dtype=mk.HIT_DTYPE)
    h["pos"] = pos
    got = _order_on_device(mk, mb, h)
    assert np.array_equal(got, _expected(h, False, np.array([3], dtype=np.int64)))
    assert mb.order_info()["path"] == 2, mb.order_info()


def test_order_library_fallback(mk):
    """a batch that defeats both binning attempts -- 20 000 tuples in record 0 and a single one in record 2^40: the
    top bits of the 59-bit key cannot separate the 20 000 -- is ordered by the library sort (and says so)"""
    rnd = random.Random(4)
    pats = _patterns(rnd, 700, [31])
    m = mk.Matcher(pats)
    plen = np.array([31] * len(pats), dtype=np.int64)
    h = _tuples(mk, rnd, 20_000, 1, 200, len(pats))
    far = np.zeros(1, dtype=mk.HIT_DTYPE)
    far["rec"], far["pat"], far["pos"] = 1 << 40, 5, 9
    h = np.concatenate([h, far])
    np.random.default_rng(1).shuffle(h)
    got = _order_on_device(mk, m, h)
    assert np.array_equal(got, _expected(h, True, plen))
    assert m.order_info()["path"] == 3, m.order_info()


def test_order_on_a_genome_like_batch(mk):
    """one 200 Mbp record (a chromosome of a genome FASTA) with one repeat k-mer every ~2 kbp and 2 000 other k-mers
    planted at random: the tuples of the real scan, ordered on the device, against the reference's comparator -- and
    WHICH path of order_hits.hip ran (record bins cannot split one record; the question is whether the re-binning on
    (record, end) holds or the library merge sort takes over: profiles/r04_order_skew.txt)"""
    torch = pytest.importorskip("torch")
    import order_skew
    r = order_skew.run(mk, torch)
    assert r["ordered_like_the_reference"] and r["tuples"] > 100_000
    assert r["order"]["path"] == 2, r  # bins on the whole (record, end, pattern) key; no library sort

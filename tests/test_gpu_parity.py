"""GPU parity tests: the gfx950 path (through the C ABI) vs the CPU oracle and the reference's
golden fixtures.  Bit-exact bar: record ids, pattern indices, 0-based positions, order,
counters, km tag strings.  Run on the GPU box with `-m gpu`.
"""
import json
import os
import random

import numpy as np
import pytest

import naive
import oracle_binding as ob
import textio

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mk():
    from merkurio_amd import native
    native.load()
    if native.device_count() < 1:
        pytest.fail("no HIP device visible: the gpu-marked tests need an MI355X")
    return native


# ------------------------------------------------------------------ reference unit tests, on the GPU
def test_bndmq_kats_gpu(mk):
    b = mk.BNDMq(b"abc", 2)  # src/pattern_matching.rs:353-363
    it = b.find_iter(b"abcabcabc")
    assert [next(it), next(it), next(it)] == [0, 3, 6] and next(it, None) is None
    assert list(mk.BNDMq(b"1234567890", 2).find_iter(b"123")) == []  # :366-373
    assert list(b.find_iter(b"")) == []  # :376-383
    assert b.find_all(b"aabcabcabc") == [1, 4, 7]  # :386-392
    assert b.find_match(b"abcabcabc") is True and b.find_match(b"defdefdef") is False  # :467-482
    with pytest.raises(mk.PatternError) as e:  # :431-437
        mk.BNDMq(b"abc", 4)
    assert e.value.kind == "InvalidQGramLength"
    with pytest.raises(mk.PatternError) as e:  # :440-446
        mk.BNDMq(b"abc", 0)
    assert e.value.kind == "InvalidQGramLength"
    with pytest.raises(mk.PatternError) as e:  # :458-464
        mk.BNDMq(b"", 1)
    assert e.value.kind == "EmptyPattern"
    with pytest.raises(mk.PatternError) as e:  # src/pattern_preprocessing.rs:80-84 via BNDMq::new
        mk.BNDMq(b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ1234567890!@#$%^&*()_+", 2)
    assert e.value.kind == "PatternTooLong"
    # 64-byte pattern is the BNDMq maximum and must work
    p64 = bytes(random.Random(5).choice(b"ACGT") for _ in range(64))
    assert mk.BNDMq(p64, 6).find_all(b"TT" + p64 + b"A" + p64) == [2, 67]


def test_algorithm_selection_gpu(mk):
    assert mk.Matcher([b"AAA", b"CCC"]).use_ac is False  # helpers.rs:555-558
    assert mk.Matcher([b"A" * 65]).use_ac is True  # :561-567
    rnd = random.Random(1)
    pats = sorted({bytes(rnd.choice(b"ACGT") for _ in range(9)) for _ in range(14)})
    assert len(pats) == 14 and mk.Matcher(pats).use_ac is True
    assert mk.Matcher(pats, q=3).use_ac is False  # -q given => BNDMq (cmd_extract.rs:169)
    assert mk.Matcher([b"ACG"], case_insensitive=True).use_ac is True  # :166-167
    assert mk.Matcher([b"ACG"], algo=mk.MK_ALGO_AC).use_ac is True


# ------------------------------------------------------------------ golden fixtures through the GPU
def _json_rows(j):
    return [(m["file"], m["record_id"], m["pattern"], int(m["position"])) for m in j["matching_records"]]


def _named_rows(rows, files, ids, patterns):
    return [(files[f], ids[f][r].decode(), patterns[p].decode(), pos) for (f, r, p, pos) in rows]


def _check_summary(j, patterns, c):
    s = j["summary_statistics"]
    assert s["number_of_records_searched"] == c["records"]
    assert s["number_of_characters_searched"] == c["bases"]
    assert s["number_of_matches"] == sum(c["hits"])
    assert s["number_of_distinct_records_with_a_hit"] == sum(c["records_hit"])
    assert s["number_of_patterns_found"] == sum(1 for x in c["pattern_hit_counts"] if x)
    assert j["pattern_hit_counts"] == {p.decode(): n for p, n in zip(patterns, c["pattern_hit_counts"])}


@pytest.mark.parametrize("name,invert", [("simple", False), ("simple-inv", True)])
def test_extract_simple_fixture_gpu(mk, golden, name, invert):
    recs = textio.read_fastx(os.path.join(golden, "fixtures/input/simple.fasta"))
    patterns = mk.parse_pattern_list(kmer_seq=["ACG"], reverse_complement=True)
    m = mk.Matcher(patterns)
    assert not m.use_ac
    keep, rows, c = m.extract_single([s for _, s in recs], logging=True, invert=invert)
    j = json.load(open(os.path.join(golden, f"fixtures/extract/{name}.json")))
    assert _named_rows(rows, ["simple.fasta"], [[i for i, _ in recs]], patterns) == _json_rows(j)
    _check_summary(j, patterns, c)
    assert j["paired_end_reads_statistics"]["number_of_extracted_records"] == c["extracted"]
    out = textio.read_fastx(os.path.join(golden, f"fixtures/extract/{name}.extracted.fasta"))
    assert [r for r, k in zip(recs, keep) if k] == out
    keep2, _, c2 = m.extract_single([s for _, s in recs], logging=False, invert=invert)
    assert keep2 == keep and c2["records"] == 0 and c2["extracted"] == c["extracted"]


def test_extract_fixed_width_fixture_gpu(mk, golden):
    recs = textio.read_fastx(os.path.join(golden, "fixtures/input/fixed-width.faa"))
    patterns = mk.parse_pattern_list(kmer_seq=["DKAT"])
    keep, rows, c = mk.Matcher(patterns).extract_single([s for _, s in recs])
    j = json.load(open(os.path.join(golden, "fixtures/extract/fixed-width.json")))
    assert _named_rows(rows, ["fixed-width.faa"], [[i for i, _ in recs]], patterns) == _json_rows(j)
    _check_summary(j, patterns, c)
    assert keep == [True]


def test_extract_paired_fixture_gpu(mk, golden):
    r1 = textio.read_fastx(os.path.join(golden, "fixtures/input/paired-1.fastq"))
    r2 = textio.read_fastx(os.path.join(golden, "fixtures/input/paired-2.fastq"))
    patterns = mk.parse_pattern_list(kmer_seq=["CTT"])
    m = mk.Matcher(patterns)
    keep, rows, c = m.extract_paired([s for _, s in r1], [s for _, s in r2])
    j = json.load(open(os.path.join(golden, "fixtures/extract/paired.json")))
    assert _named_rows(rows, ["paired-1.fastq", "paired-2.fastq"], [[i for i, _ in r1], [i for i, _ in r2]],
                       patterns) == _json_rows(j)
    _check_summary(j, patterns, c)
    p = j["paired_end_reads_statistics"]
    assert (p["number_of_hits_in_file_1"], p["number_of_hits_in_file_2"]) == c["hits"]
    assert p["number_of_extracted_records"] == c["extracted"] == 4 and keep == [True, True]
    with pytest.raises(mk.MerkurioError) as e:
        m.extract_paired([b"A"], [b"A", b"C"])
    assert e.value.code == mk.MK_E_PAIR_MISMATCH


@pytest.mark.parametrize("name,flt,inv,out", [
    ("simple", True, False, "simple.extracted.sam"),
    ("simple-inv", False, True, "simple-inv.extracted.sam"),
    ("simple-bam", False, False, "simple.tagged.extracted.sam"),
])
def test_tag_fixtures_gpu(mk, golden, name, flt, inv, out):
    hdr, recs = textio.read_sam(os.path.join(golden, "fixtures/input/simple.sam"))
    patterns = mk.parse_pattern_list(kmer_seq=["CTC"], reverse_complement=True)
    m = mk.Matcher(patterns)
    keep, rows, c, found = m.tag_records([r[9] for r in recs], logging=True, filter_matching=flt, invert=inv)
    j = json.load(open(os.path.join(golden, f"fixtures/tag/{name}.json")))
    fname = j["meta_information"]["input_files"]["record_file_1"]
    assert _named_rows(rows, [fname], [[r[0] for r in recs]], patterns) == _json_rows(j)
    _check_summary(j, patterns, c)
    _, exp = textio.read_sam(os.path.join(golden, f"fixtures/tag/{out}"))
    got = [r + [b"km:Z:" + m.tag_value(f)] for r, k, f in zip(recs, keep, found) if k]
    assert got == exp
    assert m.tag_value([1], b"ZZZ,CTC") == b"CTC,GAG,ZZZ"


def test_aho_corasick_log_json_gpu(mk, golden):
    """the only AC known-answer vector the reference holds (96 ordered hits, mixed lengths 1-3)"""
    j = json.load(open(os.path.join(golden, "fixtures/extract/log.json")))
    hdr, recs = textio.read_sam(os.path.join(golden, "fixtures/input/simple.sam"))
    patterns = mk.parse_pattern_list(kmer_seq=["CTC", "AC", "CT", "AA", "T", "A", "C", "G", "GA", "AG"],
                                     reverse_complement=True)
    m = mk.Matcher(patterns)
    assert len(patterns) == 14 and m.use_ac
    keep, rows, c, found = m.tag_records([r[9] for r in recs], logging=True)
    assert _named_rows(rows, ["simple.bam"], [[r[0] for r in recs]], patterns) == _json_rows(j)
    _check_summary(j, patterns, c)
    ac = mk.AhoCorasick(patterns)
    for r in recs:
        assert list(ac.find_overlapping_iter(r[9])) == naive.ac_order(patterns, r[9])


def test_example_minimal_gpu(mk, golden):
    recs = textio.read_fastx(os.path.join(golden, "example-minimal/sample.fasta"))
    patterns = mk.parse_pattern_list(kmer_file=os.path.join(golden, "example-minimal/kmers.txt"))
    m = mk.Matcher(patterns)
    assert patterns == [b"AAC"] and not m.use_ac
    keep, rows, c = m.extract_single([s for _, s in recs])
    assert keep == [True, True]
    assert [pos for (f, r, p, pos) in rows if r == 0] == [48, 54, 321, 450, 486, 637, 741, 849]
    assert [pos for (f, r, p, pos) in rows if r == 1] == [83, 109, 399, 451, 556, 592, 642, 752, 788, 832, 851]


def test_example_workflow_gpu(mk, golden):
    wf = os.path.join(golden, "example-workflow")
    r1 = textio.read_fastx(os.path.join(wf, "data/mutant_R1.subset.fastq.gz"))
    r2 = textio.read_fastx(os.path.join(wf, "data/mutant_R2.subset.fastq.gz"))
    patterns = mk.parse_pattern_list(kmer_file=os.path.join(wf, "significant_kmers.txt"), reverse_complement=True)
    m = mk.Matcher(patterns)
    assert len(patterns) == 6 and not m.use_ac
    keep, rows, c = m.extract_paired([s for _, s in r1], [s for _, s in r2])
    j = json.load(open(os.path.join(wf, "logs/mutant_extracted.stats.json")))
    assert _named_rows(rows, ["mutant_R1.fastq", "mutant_R2.fastq"], [[i for i, _ in r1], [i for i, _ in r2]],
                       patterns) == _json_rows(j)
    assert j["pattern_hit_counts"] == {p.decode(): n for p, n in zip(patterns, c["pattern_hit_counts"])}
    assert c["hits"] == (13, 23) and c["records_hit"] == (9, 15) and c["extracted"] == 48
    e1 = textio.read_fastx(os.path.join(wf, "output/mutant_extracted_1.fastq"))
    assert [r for r, k in zip(r1, keep) if k] == e1
    # tag golden (no logging => per-pattern find_match in the reference)
    hdr, recs = textio.read_sam(os.path.join(wf, "output/mutant_extracted.sorted.sam"))
    _, exp = textio.read_sam(os.path.join(wf, "output/mutant_extracted.sorted.tagged.sam"))
    keep, rows, c, found = m.tag_records([r[9] for r in recs], logging=False)
    assert [r + [b"km:Z:" + m.tag_value(f)] for r, f in zip(recs, found)] == exp


# ------------------------------------------------------------------ differential vs the oracle
def _rand_seq(rnd, n, alpha):
    return bytes(rnd.choice(alpha) for _ in range(n))


def _make_case(seed, n_pat, lens, n_rec, rec_len, alpha=b"ACGT", plant=0.3, ci=False):
    rnd = random.Random(seed)
    raw = [_rand_seq(rnd, rnd.choice(lens), alpha) for _ in range(n_pat)]
    recs = []
    for _ in range(n_rec):
        n = rnd.choice(rec_len) if isinstance(rec_len, (list, tuple)) else rec_len
        s = bytearray(_rand_seq(rnd, n, alpha))
        if raw and rnd.random() < plant:
            p = rnd.choice(raw)
            if ci:
                p = bytes(c ^ 0x20 if rnd.random() < 0.5 and chr(c).isalpha() else c for c in p)
            if len(p) <= n:
                k = rnd.randrange(0, n - len(p) + 1)
                s[k:k + len(p)] = p
        recs.append(bytes(s))
    return raw, recs


def _oracle_hits(patterns, use_ac, recs, ci=False):
    om = ob.Matcher(patterns, use_ac, 0, ci)
    keep, rows, c, found = ob.tag_records(om, recs, logging=True)
    return [(r, p, pos) for (_, r, p, pos) in rows], c, found


CASES = [
    # (seed, n_pat, pattern lengths, n_rec, record length(s), alphabet, algo)
    (1, 3, [31], 400, 150, b"ACGT", None),            # BNDMq domain, S=16
    (2, 40, [31], 600, 150, b"ACGT", None),           # AC, S=16
    (3, 5000, [31], 3000, 150, b"ACGT", None),        # AC, S=8
    (4, 10000, [31], 3000, 150, b"ACGT", None),       # headline geometry, S=4
    (5, 300, [21], 800, 250, b"ACGT", None),          # 21-mers
    (6, 64, [5, 9, 17, 31, 70], 500, [0, 1, 30, 149, 151, 400], b"ACGTN", None),  # mixed lengths, ragged records
    (7, 20, [1, 2, 3], 60, [0, 5, 20], b"ACGT", None),  # tiny patterns: every position hits
    (8, 12, [17], 300, 120, b"ACDEFGHIKLMNPQRSTVWY", None),  # protein alphabet
    (9, 13, [64], 200, 300, b"AC", None),             # BNDMq max length, low entropy
    (10, 200, [31], 500, 150, b"ACGTacgtN", "ci"),    # ascii_case_insensitive
    (11, 30, [33, 40, 100], 300, 200, b"ACGT", None),  # q capped at 32
    (12, 2, [8], 50, 5000, b"A", None),               # poly-A text: overlapping hits everywhere
]


@pytest.mark.parametrize("case", CASES, ids=[f"case{c[0]}" for c in CASES])
def test_scan_matches_oracle(mk, case):
    seed, n_pat, lens, n_rec, rec_len, alpha, opt = case
    ci = opt == "ci"
    raw, recs = _make_case(seed, n_pat, lens, n_rec, rec_len, alpha, ci=ci)
    patterns = mk.parse_pattern_list(kmer_seq=raw)
    rc, opats = ob.parse_pattern_list(raw)
    assert patterns == opats
    m = mk.Matcher(patterns, case_insensitive=ci)
    use_ac = ob.select_aho_corasick(ci, False, False, patterns)
    assert m.use_ac == use_ac
    exp, c_exp, found_exp = _oracle_hits(patterns, use_ac, recs, ci)
    flags, hits = m.scan(recs, mk.MK_MODE_HITS)
    got = list(zip(hits["rec"].tolist(), hits["pat"].tolist(), hits["pos"].tolist()))
    assert got == exp
    assert flags.tolist() == [bool(f) for f in found_exp]
    flags_any, _ = m.scan(recs, mk.MK_MODE_ANY)
    assert flags_any.tolist() == flags.tolist()
    # driver loops: counters with the per-matcher pattern_hit_counts semantics
    keep, rows, c, found = m.tag_records(recs, logging=True)
    assert c == c_exp and found == [sorted(set(f)) for f in found_exp]
    om = ob.Matcher(patterns, use_ac, 0, ci)
    for logging in (True, False):
        k1, r1, c1 = ob.extract_single(om, recs, logging=logging, invert=False)
        k2, r2, c2 = m.extract_single(recs, logging=logging, invert=False)
        assert k1 == k2 and r1 == r2 and c1 == c2
    half = len(recs) // 2
    k1, r1, c1 = ob.extract_paired(om, recs[:half], recs[half:2 * half], logging=True, invert=True)
    k2, r2, c2 = m.extract_paired(recs[:half], recs[half:2 * half], logging=True, invert=True)
    assert k1 == k2 and r1 == r2 and c1 == c2


@pytest.mark.parametrize("stride", [1, 2, 4, 8, 16])
def test_all_strides_agree(mk, stride):
    """every kernel variant (sampling stride) yields the same result set"""
    raw, recs = _make_case(77, 500, [31], 1500, 150)
    patterns = mk.parse_pattern_list(kmer_seq=raw, reverse_complement=True)
    m = mk.Matcher(patterns, options=dict(force_stride=stride))
    assert m.filter_info()["stride"] == stride
    exp, _, _ = _oracle_hits(patterns, True, recs)
    flags, hits = m.scan(recs)
    assert list(zip(hits["rec"].tolist(), hits["pat"].tolist(), hits["pos"].tolist())) == exp
    # narrow keys (q <= 16) too
    raw2, recs2 = _make_case(78, 100, [16 + stride - 1], 800, 100)
    p2 = mk.parse_pattern_list(kmer_seq=raw2)
    m2 = mk.Matcher(p2, options=dict(force_stride=stride))
    assert m2.filter_info()["q_gram"] == 16
    exp2, _, _ = _oracle_hits(p2, True, recs2)
    _, h2 = m2.scan(recs2)
    assert list(zip(h2["rec"].tolist(), h2["pat"].tolist(), h2["pos"].tolist())) == exp2


def test_record_boundaries_and_edges(mk):
    """occurrences that would straddle two records must not be reported; first/last positions must"""
    pat = b"ACGTTGCAACGTTGCAACGTTGCAACGTTGC"  # 31
    m = mk.Matcher([pat])
    recs = [pat[:15], pat[15:], pat, b"", pat + pat, b"G" + pat, pat[:-1], b"", pat]
    flags, hits = m.scan(recs)
    exp = [(r, p) for r, s in enumerate(recs) for p in naive.occurrences(pat, s)]
    assert exp == [(2, 0), (4, 0), (4, 31), (5, 1), (8, 0)]
    assert list(zip(hits["rec"].tolist(), hits["pos"].tolist())) == exp
    assert flags.tolist() == [bool(naive.occurrences(pat, s)) for s in recs]
    # empty batch / all-empty records
    f, h = m.scan([])
    assert len(f) == 0 and len(h) == 0
    f, h = m.scan([b"", b""])
    assert f.tolist() == [False, False] and len(h) == 0
    # capacity protocol: never truncate silently
    with pytest.raises(mk.MerkurioError) as e:
        m.scan([pat] * 10, hits_cap=3)
    assert e.value.code == mk.MK_E_CAPACITY


@pytest.mark.parametrize("k,options", [
    (31, None), (31, dict(tile_run=4)), (31, dict(tile_run=8)), (21, dict(force_stride=4)),
    (21, dict(force_global_filter=True, force_stride=8)),             # context kernel <8,14>
    (21, dict(force_global_filter=True, force_stride=4)),             # context kernel <4,18>
    (31, dict(force_global_filter=True, force_stride=8, tile_run=2)),  # context kernel <8,24>
    (21, dict(force_global_filter=True, force_stride=2)),             # global filter, runtime q
])
def test_occurrences_across_chunk_and_tile_borders(mk, k, options):
    """occurrences that start up to k bases before every 1 KiB chunk border, 31 KiB tile border and the
    start of the guarded tail are found exactly once by every kernel family (halo lanes, the packed
    16 bases in front of a tile that feed the context fingerprints, run-length tile dealing)"""
    rnd = np.random.default_rng(100 + k)
    n = 31744 * 9 + 5000  # nine main tiles + a tail
    seq = bytearray(np.frombuffer(b"ACGT", dtype=np.uint8)[rnd.integers(0, 4, n)].tobytes())
    pats = [bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[rnd.integers(0, 4, k)]) for _ in range(64)]
    offs = (0, 1, 7, 8, 9, 15, 16, 17, k - 1)  # bases of the occurrence in front of the border
    plants = [(31744 * t, offs[t - 1]) for t in range(1, 10)]                      # one tile border per offset
    plants += [(1024 * c, offs[i % 9]) for i, c in enumerate(range(2, 270, 7))]     # chunk borders, all offsets
    plants += [(31744 * 9 + 1024 * c, offs[c % 9]) for c in range(1, 4)] + [(n, k)]  # guarded tail, very end
    for i, (b, o) in enumerate(plants):
        at = b - o
        if 0 <= at <= n - k:
            seq[at:at + k] = pats[i % len(pats)]
    seq = bytes(seq)
    patterns = mk.parse_pattern_list(kmer_seq=pats)
    m = mk.Matcher(patterns, options=options)
    flags, hits = m.scan([seq[:40000], seq[40000:], seq])  # record borders inside tiles as well
    exp = []
    for r, s in enumerate((seq[:40000], seq[40000:], seq)):
        exp += [(r, p, pos) for p, pos in naive.ac_order(patterns, s)]
    assert len(exp) >= 2 * (len(plants) - 8)  # every plant shows up in two of the three records
    assert list(zip(hits["rec"].tolist(), hits["pat"].tolist(), hits["pos"].tolist())) == exp


_LENS31 = [31, 32, 33, 40, 47, 48, 49, 63, 64, 65, 100, 257]


@pytest.mark.parametrize("lens,options", [
    (_LENS31, dict(force_stride=16)), (_LENS31, dict(force_stride=8)), (_LENS31, dict(force_stride=4)),
    ([21, 23, 24, 25, 31, 32, 33, 48, 70], dict(force_stride=4)),
    ([8, 9, 15, 16, 17, 23, 24, 25, 31, 32, 33, 47, 48, 49, 65], dict(length_classes=1)),  # runtime q <= 16 (one class: two flavours)
    ([8, 9, 15, 16, 17, 31, 32, 33, 65], dict(force_stride=2)),
    ([25, 26, 31, 32, 33, 40, 48, 49, 65], dict(force_stride=4)),                 # runtime q in 17..32
    ([27, 31, 32, 33, 64, 65, 100], dict(force_stride=1)),
], ids=["16x16", "8x24", "4x28", "4x18", "rt-narrow", "rt-narrow-s2", "rt-wide-s4", "rt-wide-s1"])
def test_sparse_and_dense_kernel_variants_agree(mk, lens, options):
    """the kernel variant for hit-dense text (plain stream loads, 16-byte loads in the exact comparison) and
    the one for sparse hits give the same result set as the oracle, for pattern lengths around the 16- and
    32-byte edges of the comparison, near misses at every byte class, occurrences that end on the last byte of
    a record and of the text; every kernel family with the filter in LDS"""
    rnd = random.Random(4242)
    raw = [_rand_seq(rnd, n, b"ACGT") for n in lens for _ in range(3)]
    near = []  # near misses: one byte off at every position class of the comparison
    for p in raw[::3]:
        for k in (0, 7, 8, 15, 16, 31, 32, len(p) // 2, len(p) - 33, len(p) - 17, len(p) - 16, len(p) - 9, len(p) - 8, len(p) - 1):
            if 0 <= k < len(p):
                near.append(p[:k] + (b"A" if p[k:k + 1] != b"A" else b"C") + p[k + 1:])
    recs = []
    for i in range(1500):
        s = bytearray(_rand_seq(rnd, rnd.choice([60, 150, 300, 700]), b"ACGT"))
        for _ in range(rnd.choice([0, 1, 1, 2, 3])):
            p = rnd.choice(raw if rnd.random() < 0.7 else near)
            if len(p) <= len(s):
                k = rnd.choice([0, len(s) - len(p), rnd.randrange(0, len(s) - len(p) + 1)])
                s[k:k + len(p)] = p
        recs.append(bytes(s))
    patterns = mk.parse_pattern_list(kmer_seq=raw)
    exp, c_exp, found_exp = _oracle_hits(patterns, True, recs)
    assert len(exp) > 600
    m = mk.Matcher(patterns, algo=mk.MK_ALGO_AC, options=options)
    names = set()
    for density in (0, 1000):
        for mode in (mk.MK_MODE_HITS, mk.MK_MODE_ANY):
            m.hint_hit_density(density)  # mk_scan_batch replaces it with what the batch showed: set before each scan
            flags, hits = m.scan(recs, mode, hits_cap=len(exp) + 16)  # one launch: no capacity retry
            names.add(m.kernel_name)
            assert flags.tolist() == [bool(f) for f in found_exp], (density, mode)
            if mode == mk.MK_MODE_HITS:
                assert list(zip(hits["rec"].tolist(), hits["pat"].tolist(), hits["pos"].tolist())) == exp, density
    assert len(names) == 4 and sum(n.endswith("plain>") for n in names) == 2, names


@pytest.mark.parametrize("L,k", [(150, 31), (100, 21), (31, 31), (33, 31), (250, 21)])
def test_fixed_record_length_batches(mk, L, k):
    """records of one length: mk_scan_batch finds that out from its offsets, device callers state it with
    mk_matcher_set_fixed_record_length (no offsets array at all) -- the record of an occurrence is then computed,
    not looked up.  Same result set as the oracle in every kernel flavour, for occurrences on the first and the
    last byte of a record and text that matches across a record border (no occurrence); a wrong byte count is
    refused."""
    torch = pytest.importorskip("torch")
    rnd = random.Random(L * 1000 + k)
    raw = [_rand_seq(rnd, k, b"ACGT") for _ in range(200)]
    patterns = mk.parse_pattern_list(kmer_seq=raw)
    n_rec = 30_000
    buf = bytearray(_rand_seq(rnd, n_rec * L, b"ACGT"))
    for i in range(0, n_rec, 3):
        p = rnd.choice(patterns)
        where = rnd.choice(["first", "last", "any", "straddle"])
        if where == "first":
            o = i * L
        elif where == "last":
            o = (i + 1) * L - k
        elif where == "any":
            o = i * L + rnd.randrange(0, L - k + 1)
        else:  # across the border to the next record: not an occurrence of either
            o = (i + 1) * L - rnd.randrange(1, k)
        if o + k <= len(buf):
            buf[o:o + k] = p
    recs = [bytes(buf[i * L:(i + 1) * L]) for i in range(n_rec)]
    exp, _, found_exp = _oracle_hits(patterns, True, recs)
    assert len(exp) > 5000
    m = mk.Matcher(patterns, algo=mk.MK_ALGO_AC)
    lib = mk.load()
    dev = torch.device("cuda", 0)
    d_seq = torch.zeros(len(buf) + 64, dtype=torch.uint8, device=dev)
    d_seq[:len(buf)] = torch.frombuffer(bytes(buf), dtype=torch.uint8).to(dev)
    d_flags = torch.zeros(n_rec + 8, dtype=torch.uint8, device=dev)
    cap = len(exp) + 16
    d_hits = torch.zeros(2 * cap, dtype=torch.int64, device=dev)
    d_nh = torch.zeros(1, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    assert lib.mk_matcher_set_fixed_record_length(m.handle, L) == 0
    names = set()
    for density in (0, 1000):
        m.hint_hit_density(density)
        # no offsets array: d_seq_off = NULL
        assert lib.mk_scan_device(m.handle, d_seq.data_ptr(), len(buf), None, n_rec, mk.MK_MODE_HITS, d_flags.data_ptr(),
                                  d_hits.data_ptr(), cap, d_nh.data_ptr(), None, st) == 0, lib.mk_last_error()
        torch.cuda.synchronize()
        names.add(m.kernel_name)
        nh = int(d_nh.item())
        assert nh == len(exp), density
        assert lib.mk_order_hits_device(m.handle, d_hits.data_ptr(), nh, st) == 0
        torch.cuda.synchronize()
        got = np.frombuffer(d_hits.cpu().numpy().tobytes(), dtype=mk.HIT_DTYPE)[:nh]
        assert list(zip(got["rec"].tolist(), got["pat"].tolist(), got["pos"].tolist())) == exp, density
        assert d_flags[:n_rec].cpu().numpy().astype(bool).tolist() == [bool(f) for f in found_exp], density
    assert len(names) == 2, names
    # the byte count must be n_rec * L
    assert lib.mk_scan_device(m.handle, d_seq.data_ptr(), len(buf) - 1, None, n_rec, mk.MK_MODE_ANY, d_flags.data_ptr(),
                              None, 0, d_nh.data_ptr(), None, st) == mk.MK_E_INVALID_ARG
    # back to offsets: a null offsets pointer is refused again
    assert lib.mk_matcher_set_fixed_record_length(m.handle, 0) == 0
    assert lib.mk_scan_device(m.handle, d_seq.data_ptr(), len(buf), None, n_rec, mk.MK_MODE_ANY, d_flags.data_ptr(),
                              None, 0, d_nh.data_ptr(), None, st) == mk.MK_E_INVALID_ARG
    # and mk_scan_batch (host buffers, equal lengths detected) agrees
    flags, hits = m.scan(recs, mk.MK_MODE_HITS, hits_cap=cap)
    assert list(zip(hits["rec"].tolist(), hits["pat"].tolist(), hits["pos"].tolist())) == exp


def test_device_out_of_memory_is_an_error_code(mk):
    """hipErrorOutOfMemory comes back across the C ABI as MK_E_NOMEM (not MK_E_HIP, not an abort), and the
    handle stays usable"""
    import ctypes as C
    pat = b"ACGTTGCAACGTTGCAACGTTGCAACGTTGC"
    m = mk.Matcher([pat])
    data, off = mk.pack_records([pat, b"ACGT"])
    flags = np.zeros(2, dtype=np.uint8)
    hits = np.zeros(4, dtype=mk.HIT_DTYPE)  # never written: the device buffer of 2^40 tuples cannot be allocated
    nh = C.c_uint64()
    rc = mk.load().mk_scan_batch(m.handle, data.ctypes.data, off.ctypes.data, 2, mk.MK_MODE_HITS, flags.ctypes.data,
                                 hits.ctypes.data, 1 << 40, C.byref(nh))
    assert rc == mk.MK_E_NOMEM, (rc, mk.load().mk_last_error())
    f, h = m.scan([pat, b"ACGT"])
    assert f.tolist() == [True, False] and h["pos"].tolist() == [0]


def test_long_single_record(mk):
    """one chromosome-sized record is split across lanes / tiles with a halo"""
    rnd = np.random.default_rng(3)
    n = 3_000_000
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rnd.integers(0, 4, n)].tobytes()
    pats = sorted({seq[i:i + 31] for i in rnd.integers(0, n - 31, 40).tolist()} | {seq[:31], seq[-31:]})
    m = mk.Matcher(pats)
    flags, hits = m.scan([seq])
    exp = naive.ac_order(pats, seq)
    assert list(zip(hits["pat"].tolist(), hits["pos"].tolist())) == exp and flags.tolist() == [True]


def test_device_counters_and_synth(mk):
    """device-resident scan: synthetic reads generated on the GPU, counters accumulated on the
    GPU, checked against the oracle run on the identical host-generated bytes"""
    import ctypes as C
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    rnd = random.Random(11)
    raw = [_rand_seq(rnd, 31, b"ACGT") for _ in range(2000)]
    patterns = mk.parse_pattern_list(kmer_seq=raw, reverse_complement=True)
    m = mk.Matcher(patterns)
    n_rec, L, seed = 200_000, 150, 12345
    d_seq = torch.empty(n_rec * L + 64, dtype=torch.uint8, device=dev)
    d_off = torch.empty(n_rec + 1, dtype=torch.int64, device=dev)
    lib = mk.load()
    st = torch.cuda.current_stream().cuda_stream
    assert lib.mk_synth_reads_device(m.handle, seed, n_rec, L, 50, d_seq.data_ptr(), d_off.data_ptr(), st) == 0
    h_seq = np.zeros(n_rec * L, dtype=np.uint8)
    h_off = np.zeros(n_rec + 1, dtype=np.uint64)
    assert lib.mk_synth_reads_host(m.handle, seed, 0, n_rec, L, 50, h_seq.ctypes.data, h_off.ctypes.data) == 0
    torch.cuda.synchronize()
    assert np.array_equal(d_seq[:n_rec * L].cpu().numpy(), h_seq)
    assert np.array_equal(d_off.cpu().numpy().astype(np.uint64), h_off)
    d_flags = torch.zeros(n_rec + 8, dtype=torch.uint8, device=dev)
    cap = 1 << 16
    d_hits = torch.zeros(cap * 2, dtype=torch.int64, device=dev)
    d_nh = torch.zeros(1, dtype=torch.int64, device=dev)
    d_cnt = torch.zeros(len(patterns) + mk.MK_NUM_SUMMARY, dtype=torch.int64, device=dev)
    rc = lib.mk_scan_device(m.handle, d_seq.data_ptr(), n_rec * L, d_off.data_ptr(), n_rec, mk.MK_MODE_HITS,
                            d_flags.data_ptr(), d_hits.data_ptr(), cap, d_nh.data_ptr(), d_cnt.data_ptr(), st)
    assert rc == 0, lib.mk_last_error()
    torch.cuda.synchronize()
    nh = int(d_nh.item())
    hits = np.frombuffer(d_hits.cpu().numpy().tobytes(), dtype=mk.HIT_DTYPE)[:nh].copy()
    lib.mk_order_hits(m.handle, hits.ctypes.data, nh)
    recs = [h_seq[i * L:(i + 1) * L].tobytes() for i in range(n_rec)]
    exp, c_exp, found = _oracle_hits(patterns, True, recs)
    assert nh == len(exp) and nh >= n_rec // 50 // 2
    assert list(zip(hits["rec"].tolist(), hits["pat"].tolist(), hits["pos"].tolist())) == exp
    cnt = d_cnt.cpu().numpy()
    assert cnt[:len(patterns)].tolist() == c_exp["pattern_hit_counts"]
    s = cnt[len(patterns):]
    assert s[mk.MK_SUM_HITS] == nh and s[mk.MK_SUM_RECORDS_HIT] == c_exp["records_hit"][0]
    assert s[mk.MK_SUM_RECORDS] == n_rec and s[mk.MK_SUM_BASES] == n_rec * L
    assert d_flags[:n_rec].cpu().numpy().astype(bool).tolist() == [bool(f) for f in found]


@pytest.mark.parametrize("algo", ["ac", "bndmq"])
def test_emission_order_on_the_device(mk, algo):
    """mk_order_hits_device (what mk_scan_batch uses from 4096 tuples up) == mk_order_hits on the host == the
    oracle's emission order: nested patterns that end on the same byte (AC: longer first), the same pattern
    set pattern-major under BNDMq, tens of thousands of tuples over short and long records"""
    torch = pytest.importorskip("torch")
    rnd = random.Random(77)
    core = _rand_seq(rnd, 40, b"ACGT")
    # suffixes / infixes of one another: many ties on the end position
    raw = [core, core[8:], core[20:], core[30:], core[5:25], core[5:17], b"ACGT", b"CGTA", b"GT"] + \
          [_rand_seq(rnd, n, b"ACGT") for n in (3, 4, 5, 9, 21, 31)]
    if algo == "bndmq":
        raw = raw[:13]  # BNDMq domain: < 14 patterns, each <= 64 bytes
    patterns = mk.parse_pattern_list(kmer_seq=raw)
    m = mk.Matcher(patterns, algo=mk.MK_ALGO_AC if algo == "ac" else mk.MK_ALGO_BNDMQ)
    recs = []
    for i in range(4000):
        s = bytearray(_rand_seq(rnd, rnd.choice([0, 3, 50, 150, 400]), b"ACGT"))
        if len(s) >= 40 and i % 3 == 0:
            k = rnd.randrange(0, len(s) - 40 + 1)
            s[k:k + 40] = core
        recs.append(bytes(s))
    exp, _, _ = _oracle_hits(patterns, algo == "ac", recs)
    assert len(exp) > 30_000
    # through mk_scan_batch (device sort inside)
    flags, hits = m.scan(recs, mk.MK_MODE_HITS, hits_cap=len(exp) + 16)
    assert list(zip(hits["rec"].tolist(), hits["pat"].tolist(), hits["pos"].tolist())) == exp
    # the two entry points on the same unordered tuples
    lib = mk.load()
    dev = torch.device("cuda", 0)
    data, off = mk.pack_records(recs)
    n_bytes = int(off[-1])
    d_seq = torch.zeros(n_bytes + 64, dtype=torch.uint8, device=dev)
    d_seq[:n_bytes] = torch.from_numpy(data[:n_bytes].copy()).to(dev)
    d_off = torch.from_numpy(off.astype(np.int64)).to(dev)
    d_flags = torch.zeros(len(recs) + 8, dtype=torch.uint8, device=dev)
    cap = len(exp) + 16
    d_hits = torch.zeros(2 * cap, dtype=torch.int64, device=dev)
    d_nh = torch.zeros(1, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    assert lib.mk_scan_device(m.handle, d_seq.data_ptr(), n_bytes, d_off.data_ptr(), len(recs), mk.MK_MODE_HITS, d_flags.data_ptr(),
                              d_hits.data_ptr(), cap, d_nh.data_ptr(), None, st) == 0, lib.mk_last_error()
    torch.cuda.synchronize()
    nh = int(d_nh.item())
    assert nh == len(exp)
    host = np.frombuffer(d_hits.cpu().numpy().tobytes(), dtype=mk.HIT_DTYPE)[:nh].copy()
    assert lib.mk_order_hits(m.handle, host.ctypes.data, nh) == 0
    assert lib.mk_order_hits_device(m.handle, d_hits.data_ptr(), nh, st) == 0, lib.mk_last_error()
    torch.cuda.synchronize()
    devs = np.frombuffer(d_hits.cpu().numpy().tobytes(), dtype=mk.HIT_DTYPE)[:nh]
    assert np.array_equal(devs, host)
    assert list(zip(host["rec"].tolist(), host["pat"].tolist(), host["pos"].tolist())) == exp
    # degenerate sizes and arguments
    assert lib.mk_order_hits_device(m.handle, d_hits.data_ptr(), 0, st) == 0
    assert lib.mk_order_hits_device(m.handle, d_hits.data_ptr(), 1, st) == 0
    assert lib.mk_order_hits_device(m.handle, None, 5, st) == mk.MK_E_INVALID_ARG
    assert lib.mk_order_hits_device(None, d_hits.data_ptr(), 5, st) == mk.MK_E_INVALID_ARG


def test_flag_clear_respects_alignment_and_bounds(mk):
    """every scan starts by clearing d_rec_flags[0, n_rec rounded up to 4) and *d_n_hits in one kernel: any
    4-byte-aligned flag pointer, any record count; bytes in front of and behind that range stay untouched"""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    lib = mk.load()
    pat = b"ACGTTGCAACGTTGCAACGTTGCAACGTTGC"
    m = mk.Matcher([pat])
    st = torch.cuda.current_stream().cuda_stream
    base = torch.empty(4096, dtype=torch.uint8, device=dev)
    d_nh = torch.empty(1, dtype=torch.int64, device=dev)
    for shift in (0, 4, 8, 12, 20):
        for n_rec in (1, 2, 3, 4, 5, 15, 16, 17, 31, 33, 64, 67, 1000):
            recs = [pat if i % 3 == 0 else b"G" * 40 for i in range(n_rec)]
            data, off = mk.pack_records(recs)
            n_bytes = int(off[-1])
            d_seq = torch.zeros(n_bytes + 64, dtype=torch.uint8, device=dev)
            d_seq[:n_bytes] = torch.from_numpy(data[:n_bytes].copy()).to(dev)
            d_off = torch.from_numpy(off.astype(np.int64)).to(dev)
            base.fill_(0xEE)
            d_nh.fill_(-1)
            flags = base[64 + shift:]
            assert lib.mk_scan_device(m.handle, d_seq.data_ptr(), n_bytes, d_off.data_ptr(), n_rec, mk.MK_MODE_ANY, flags.data_ptr(),
                                      None, 0, d_nh.data_ptr(), None, st) == 0, lib.mk_last_error()
            torch.cuda.synchronize()
            h = base.cpu().numpy()
            lo, hi = 64 + shift, 64 + shift + (n_rec + 3) // 4 * 4
            assert h[lo:lo + n_rec].tolist() == [1 if i % 3 == 0 else 0 for i in range(n_rec)], (shift, n_rec)
            assert not h[lo + n_rec:hi].any() and (h[:lo] == 0xEE).all() and (h[hi:] == 0xEE).all(), (shift, n_rec)
            assert int(d_nh.item()) == 0
    assert lib.mk_scan_device(m.handle, d_seq.data_ptr(), n_bytes, d_off.data_ptr(), n_rec, mk.MK_MODE_ANY, base[66:].data_ptr(),
                              None, 0, d_nh.data_ptr(), None, st) == mk.MK_E_INVALID_ARG

"""Length classes (matcher.cpp: plan_classes; scan_kernel_impl.hpp: MC kernels) against the CPU oracle.

The reference's Aho-Corasick DFA scans any pattern list at one speed (src/cmd_extract.rs:260-265, the K3 row of
SURVEY.md §8); here a set whose shortest pattern is much shorter than the rest is split into a main class (hashed
q-gram filter) and a short class (own stride, q-grams of <= 8 bases in a byte / bit table), both probed in one pass.
Every split must give the oracle's result set and order, bit-exact.  Run on the GPU box with `-m gpu`.
"""
import random

import numpy as np
import pytest

import naive
import oracle_binding as ob

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mk():
    from merkurio_amd import native
    native.load()
    if native.device_count() < 1:
        pytest.fail("no HIP device visible: the gpu-marked tests need an MI355X")
    return native


def _rand_seq(rnd, n, alpha=b"ACGT"):
    return bytes(rnd.choice(alpha) for _ in range(n))


def _oracle(patterns, use_ac, recs, ci=False):
    om = ob.Matcher(patterns, use_ac, 0, ci)
    keep, rows, c, found = ob.tag_records(om, recs, logging=True)
    return [(r, p, pos) for (_, r, p, pos) in rows], c, found


def _tuples(hits):
    return list(zip(hits["rec"].tolist(), hits["pat"].tolist(), hits["pos"].tolist()))


def _records(rnd, raw, n_rec, alpha=b"ACGT", lens=(60, 150, 300, 700)):
    """random records with occurrences of patterns of every length planted on the first / last byte of a record and
    anywhere, plus near misses (one byte off at the first, a middle and the last position)"""
    near = []
    for p in raw:
        for k in {0, len(p) // 2, len(p) - 1}:
            near.append(p[:k] + (b"A" if p[k:k + 1] != b"A" else b"C") + p[k + 1:])
    recs = []
    for _ in range(n_rec):
        s = bytearray(_rand_seq(rnd, rnd.choice(lens), alpha))
        for _ in range(rnd.choice([0, 1, 1, 2, 3])):
            p = rnd.choice(raw if rnd.random() < 0.7 else near)
            if len(p) <= len(s):
                k = rnd.choice([0, len(s) - len(p), rnd.randrange(0, len(s) - len(p) + 1)])
                s[k:k + len(p)] = p
        recs.append(bytes(s))
    return recs


def _family(info, global_filter=False):
    """the kernel family a main-class geometry runs with (scan_kernel.hip: launch_scan)"""
    S, q = info["stride"], info["q_gram"]
    qc = q if (S, q) in ((16, 16), (8, 24), (4, 28), (4, 18)) and not global_filter else (0 if q <= 16 else -1)
    return f"<{S},{qc},"


# (main lengths, patterns per main length, short lengths, patterns per short length, options, main-class kernel family)
SPLITS = [
    ([31], 40, [8], 1, None, "<16,16,"),                                             # rule: S2 = 4, q2 = 5, byte table
    ([31], 40, [8], 1, dict(length_classes=2, force_stride2=1), "<16,16,"),          # q2 = 8: bit table, every base a sample
    ([31], 40, [8], 2, dict(length_classes=2, force_stride2=2), "<16,16,"),          # q2 = 7: bit table
    ([31], 40, [8], 2, dict(length_classes=2, force_stride2=2, force_q2=6), "<16,16,"),  # byte table at its largest q
    ([31], 400, [9, 12], 2, dict(length_classes=2, force_stride=8), "<8,24,"),       # headline family + S2 = 4
    ([31], 400, [15, 16], 3, dict(length_classes=2, force_stride=8, force_split_len=31, force_stride2=8), "<8,24,"),  # S2 = 8, q2 = 8
    ([31, 33, 40], 30, [3, 5], 2, dict(length_classes=2, force_stride=4), "<4,28,"),  # S2 <= 2 (shortest pattern 3)
    ([21], 60, [7], 2, dict(length_classes=2, force_stride=4), "<4,18,"),
    ([16, 17, 20], 30, [6, 8], 2, dict(length_classes=2, force_split_len=16), "<8,0,"),     # main class runtime q <= 16
    ([16, 17, 20], 30, [1, 2], 1, dict(length_classes=2, force_split_len=16, force_stride=2), "<2,0,"),  # one-base patterns: S2 = 1, q2 = 1
    ([40, 48, 65, 100], 20, [10, 11], 2, dict(length_classes=2, force_stride=8), "<8,-1,"),  # main class runtime q in 17..32
    ([40, 48, 65, 100], 20, [4], 3, dict(length_classes=2, force_stride=2), "<2,-1,"),
    ([34, 64], 20, [12], 2, dict(length_classes=2), "<16,-1,"),
    # main filter in GLOBAL memory (what a set of hundreds of thousands of patterns gets): runtime-q kernels
    ([31], 400, [9, 12], 2, dict(force_global_filter=True), "<16,0,"),
    ([21], 300, [8], 1, dict(force_global_filter=True, force_stride=8, length_classes=2), "<8,0,"),
    ([40, 65], 100, [5, 14], 2, dict(force_global_filter=True, force_stride=4, length_classes=2), "<4,-1,"),
]


@pytest.mark.parametrize("case", SPLITS, ids=[f"split{i}" for i in range(len(SPLITS))])
def test_length_classes_match_oracle(mk, case):
    main_lens, n_main, short_lens, n_short, options, family = case
    rnd = random.Random(hash((tuple(main_lens), tuple(short_lens), n_main)) & 0xFFFF)
    raw = [_rand_seq(rnd, n) for n in main_lens for _ in range(n_main)]
    raw_short = [_rand_seq(rnd, n) for n in short_lens for _ in range(n_short)]
    patterns = mk.parse_pattern_list(kmer_seq=raw + raw_short)
    recs = _records(rnd, raw[::max(1, len(raw) // 40)] + raw_short, 1500)
    # an empty record, records shorter than every pattern, a record that is exactly one short pattern
    recs += [b"", b"A", raw_short[0], raw_short[-1] + raw_short[0]]
    exp, c_exp, found_exp = _oracle(patterns, True, recs)
    m = mk.Matcher(patterns, algo=mk.MK_ALGO_AC, options=options)
    ci = m.class_info()
    assert ci["split_len"] == min(main_lens) and ci["n_short"] == len({p for p in raw_short}), ci
    for mode in (mk.MK_MODE_HITS, mk.MK_MODE_ANY):
        flags, hits = m.scan(recs, mode, hits_cap=len(exp) + 16)
        gf = bool(options and options.get("force_global_filter"))
        assert m.filter_mode()["in_lds"] is (not gf)
        assert family == _family(m.filter_info(), gf) and family in m.kernel_name and m.kernel_name.endswith("2-class>"), m.kernel_name
        assert flags.tolist() == [bool(f) for f in found_exp], mode
        if mode == mk.MK_MODE_HITS:
            assert _tuples(hits) == exp
    # records of ONE length: mk_scan_batch sees that from its offsets and the kernel computes the record of an occurrence
    # instead of looking it up (the fixed-length path of the two-class kernels)
    eq = [r[:60] for r in recs if len(r) >= 60]
    exp_eq, _, found_eq = _oracle(patterns, True, eq)
    flags_eq, hits_eq = m.scan(eq, mk.MK_MODE_HITS, hits_cap=len(exp_eq) + 16)
    assert _tuples(hits_eq) == exp_eq and flags_eq.tolist() == [bool(f) for f in found_eq]
    # the same set as ONE class: same answer (what round 3 computed, at the stride the shortest pattern dictates)
    one = mk.Matcher(patterns, algo=mk.MK_ALGO_AC, options=dict(length_classes=1))
    assert one.class_info()["split_len"] == 0
    _, hits1 = one.scan(recs, mk.MK_MODE_HITS, hits_cap=len(exp) + 16)
    assert _tuples(hits1) == exp
    # driver loops on top of the two-class scan: counters, pattern sets, BNDMq-order rows
    keep, rows, c, found = m.tag_records(recs, logging=True)
    assert c == c_exp and found == [sorted(set(f)) for f in found_exp]


@pytest.mark.parametrize("s2,q2", [(4, 0), (1, 8), (2, 6), (8, 0)])
def test_short_class_across_chunk_and_tile_borders(mk, s2, q2):
    """occurrences of short-class patterns that start up to their length before every 1 KiB chunk border, 31 KiB tile
    border and the guarded tail are found exactly once (halo lanes of the short class's samples)"""
    k_short = 15 if s2 == 8 else 8
    rnd = np.random.default_rng(500 + s2)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    n = 31744 * 9 + 5000
    seq = bytearray(acgt[rnd.integers(0, 4, n)].tobytes())
    longs = [bytes(acgt[rnd.integers(0, 4, 31)]) for _ in range(32)]
    shorts = [bytes(acgt[rnd.integers(0, 4, k_short)]) for _ in range(3)]
    offs = list(range(0, k_short)) + [k_short]
    plants = [(31744 * t, offs[t % len(offs)]) for t in range(1, 10)]
    plants += [(1024 * c, offs[i % len(offs)]) for i, c in enumerate(range(2, 270, 5))]
    plants += [(31744 * 9 + 1024 * c, offs[c % len(offs)]) for c in range(1, 4)] + [(n, k_short)]
    for i, (b, o) in enumerate(plants):
        at = b - o
        if 0 <= at <= n - k_short:
            seq[at:at + k_short] = shorts[i % 3]
    for i in range(40):  # the main class keeps working next to it
        at = int(rnd.integers(0, n - 31))
        seq[at:at + 31] = longs[i % 32]
    seq = bytes(seq)
    patterns = mk.parse_pattern_list(kmer_seq=longs + shorts)
    m = mk.Matcher(patterns, options=dict(length_classes=2, force_stride2=s2, force_q2=q2))
    ci = m.class_info()
    assert ci["stride2"] == s2 and ci["n_short"] == 3, ci
    parts = (seq[:40000], seq[40000:], seq)
    flags, hits = m.scan(list(parts))
    assert m.kernel_name.endswith("2-class>")
    exp = []
    for r, s in enumerate(parts):
        exp += [(r, p, pos) for p, pos in naive.ac_order(patterns, s)]
    assert len(exp) >= 2 * (len(plants) - 8)
    assert _tuples(hits) == exp


def test_classes_with_case_insensitive_and_other_alphabets(mk):
    """-I (ASCII case folding) and bytes outside ACGT: the 2-bit codes only feed the filters, level 3 decides"""
    rnd = random.Random(99)
    alpha = b"ACGTacgtNRY"
    raw = [_rand_seq(rnd, 31, alpha) for _ in range(60)] + [_rand_seq(rnd, 7, alpha) for _ in range(3)]
    patterns = mk.parse_pattern_list(kmer_seq=raw)
    recs = _records(rnd, raw[::4] + raw[-3:], 800, alpha=alpha)
    # the same text with the case of every other letter flipped
    recs += [bytes(c ^ 0x20 if (i & 1) and chr(c).isalpha() else c for i, c in enumerate(r)) for r in recs[:200]]
    for ci in (False, True):
        exp, c_exp, found_exp = _oracle(patterns, True, recs, ci)
        m = mk.Matcher(patterns, algo=mk.MK_ALGO_AC, case_insensitive=ci, options=dict(length_classes=2))
        flags, hits = m.scan(recs, mk.MK_MODE_HITS, hits_cap=len(exp) + 16)
        assert m.kernel_name.endswith("2-class>")
        assert _tuples(hits) == exp and flags.tolist() == [bool(f) for f in found_exp]


def test_class_rule(mk):
    """what the cost model decides: the headline set + one 8-mer keeps the headline's geometry for its 31-mers; a
    uniform set, a set forced to one stride and a set too large for the LDS filter stay one class; a set of
    lengths 15..31 is one class at the stride its q-gram floor admits"""
    rnd = random.Random(7)
    k31 = [_rand_seq(rnd, 31) for _ in range(10_000)]
    p = mk.parse_pattern_list(kmer_seq=k31 + [b"GATTACAG"])
    m = mk.Matcher(p)
    assert m.class_info() == {"split_len": 31, "n_short": 1, "q_gram2": 5, "stride2": 4}
    assert (m.filter_info()["stride"], m.filter_info()["q_gram"]) == (8, 24)
    assert mk.Matcher(mk.parse_pattern_list(kmer_seq=k31)).class_info()["split_len"] == 0
    assert mk.Matcher(p, options=dict(force_stride=1)).class_info()["split_len"] == 0
    mixed = mk.parse_pattern_list(kmer_seq=[s[:rnd.randrange(15, 32)] for s in k31])
    mm = mk.Matcher(mixed)
    assert mm.class_info()["split_len"] == 0 and (mm.filter_info()["stride"], mm.filter_info()["q_gram"]) == (4, 12)
    with pytest.raises(mk.MerkurioError):
        mk.Matcher(mk.parse_pattern_list(kmer_seq=k31), options=dict(length_classes=2))  # nothing to split
    # a hundred 10-mers next to the 31-mers: still worth a class of their own (bit table, q2 = 7 or 8)
    p100 = mk.parse_pattern_list(kmer_seq=k31 + [_rand_seq(rnd, 10) for _ in range(100)])
    ci = mk.Matcher(p100).class_info()
    assert ci["split_len"] == 31 and ci["n_short"] == 100 and ci["q_gram2"] >= 7, ci

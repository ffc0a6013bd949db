"""Pins the CPU oracle (oracle/mk_oracle.c) to every known-answer vector the reference
holds for the pattern-matching hot path.  CPU only.

KAT sources (reference file:line):
  src/pattern_preprocessing.rs:54-84, src/pattern_matching.rs:353-488,
  src/helpers.rs:277-431,555-567, tests/fixtures/**, example-minimal, example-workflow.
"""
import json
import os
import random

import pytest

import naive
import oracle_binding as ob
import textio


# ------------------------------------------------------------------ unit KATs
def test_generate_masks_kats():
    rc, masks, accept = ob.generate_masks(b"abc")  # pattern_preprocessing.rs:54-68
    assert rc == 0 and accept == 4
    assert (masks[97], masks[98], masks[99]) == (4, 2, 1)
    rc, masks, accept = ob.generate_masks(b"3$$X3")  # :70-77
    assert (masks[51], masks[36], masks[88], accept) == (17, 12, 2, 16)
    long = b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ1234567890!@#$%^&*()_+"
    assert ob.generate_masks(long)[0] == ob.MKO_E_PATTERN_TOO_LONG  # :80-84
    rc, masks, accept = ob.generate_masks(b"A" * 64)
    assert rc == 0 and accept == 1 << 63 and masks[65] == (1 << 64) - 1


def test_bndmq_kats():
    b = ob.BNDMq(b"abc", 2)  # pattern_matching.rs:353-363
    assert b.find_all(b"abcabcabc") == [0, 3, 6]
    assert ob.BNDMq(b"1234567890", 2).find_all(b"123") == []  # :366-373
    assert b.find_all(b"") == []  # :376-383
    assert b.find_all(b"aabcabcabc") == [1, 4, 7]  # :386-392
    assert ob.bndm_find_all(b"abc", b"abcabcabc") == [0, 3, 6]  # :395-401
    assert ob.bndm_find_all(b"1234567890", b"123") == []
    assert ob.BNDMq(b"abc", 4).rc == ob.MKO_E_INVALID_Q  # :431-437
    assert ob.BNDMq(b"abc", 0).rc == ob.MKO_E_INVALID_Q  # :440-446
    assert ob.BNDMq(b"", 1).rc == ob.MKO_E_EMPTY_PATTERN  # :458-464
    assert b.find_match(b"abcabcabc") is True  # :467-473
    assert b.find_match(b"defdefdef") is False  # :476-482
    assert ob.tune_q_value(len("AAAAAAAACCCCCCCCGGGGGGGGTTTTTTT")) == 5  # :485-488
    assert [ob.tune_q_value(n) for n in (0, 1, 2, 3, 4, 8, 9, 30, 31, 55, 56, 64, 65)] == \
        [1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 0]


def test_recommend_aho_corasick_kats():
    assert ob.recommend_aho_corasick([b"AAA", b"CCC"]) is False  # helpers.rs:555-558
    assert ob.recommend_aho_corasick([b"AAAAAAAACCCCCCCCGGGGGGGGTTTTTTTTAAAAAAAACCCCCCCCGGGGGGGGTTTTTTTTA"]) is True
    assert ob.recommend_aho_corasick([b"A"] * 14) is True and ob.recommend_aho_corasick([b"A"] * 13) is False
    # cmd_extract.rs:166-171
    assert ob.select_aho_corasick(True, False, True, [b"A"]) is True
    assert ob.select_aho_corasick(False, False, True, [b"A"] * 20) is False  # -q forces BNDMq
    assert ob.select_aho_corasick(False, True, False, [b"A"]) is True


def _kmers(golden, name):
    rc, k = ob.read_kmers_from_text(open(os.path.join(golden, "data", name), "rb").read())
    return rc, k


def test_read_kmers_kats(golden):
    three = {b"AAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAA", b"AAAATTGCATGAATATTGTAGATCAAAGCACA",
             b"CTCCGAAGAAGTTGCTGTTCTTGATGGTTATT"}
    for name in ("kmers.txt", "kmers.fasta"):  # helpers.rs:277-297
        rc, k = _kmers(golden, name)
        assert rc == 0 and len(k) == 3 and set(k) == three
    rc, k = _kmers(golden, "kmers-messy.txt")  # :300-309
    assert rc == 0 and k == [b"AAAAAAAAAAAAAAAAAAAAAAAAAAAA", b"TTGCATGAATATTGTA", b"CTCCGAAGAAGTTGCTGTTCTTGATGGTTATT"]
    # a whitespace-only line survives the filter and trims to "" (helpers.rs:152-156)
    assert ob.read_kmers_from_text(b"AC\n   \n#x\n>y\n  #notcomment\nGT\r\n")[1] == [b"AC", b"", b"#notcomment", b"GT"]
    assert _kmers(golden, "kmers-empty.txt")[0] == ob.MKO_E_NO_PATTERNS  # :312-316


def test_parse_pattern_list_kats(golden):
    rc, raw = _kmers(golden, "kmers-duplicates.txt")
    rc, pl = ob.parse_pattern_list(raw, reverse_complement=True)  # helpers.rs:319-331
    assert rc == 0 and len(pl) == 4
    rc, raw = _kmers(golden, "kmers.txt")
    rc, pl = ob.parse_pattern_list(raw)  # :334-348
    assert pl == [b"AAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAA", b"AAAATTGCATGAATATTGTAGATCAAAGCACA",
                  b"CTCCGAAGAAGTTGCTGTTCTTGATGGTTATT"]
    assert ob.parse_pattern_list([b""])[0] == ob.MKO_E_NO_PATTERNS  # :351-361
    rc, pl = ob.parse_pattern_list(raw, canonical=True)  # :364-378
    assert set(pl) == {b"AATAACCATCAAGAACAGCAACTTCTTCGGAG", b"AAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAA",
                       b"AAAATTGCATGAATATTGTAGATCAAAGCACA"}
    rc, pl = ob.parse_pattern_list(raw, reverse_complement=True)  # :381-398
    assert len(pl) == 6 and set(pl) == {
        b"AAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAA", b"AAAATTGCATGAATATTGTAGATCAAAGCACA", b"CTCCGAAGAAGTTGCTGTTCTTGATGGTTATT",
        b"TTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTT", b"TGTGCTTTGATCTACAATATTCATGCAATTTT", b"AATAACCATCAAGAACAGCAACTTCTTCGGAG"}
    assert pl == sorted(pl)
    rc, raw_aa = _kmers(golden, "kmers-aa.txt")
    rc, pl = ob.parse_pattern_list(raw_aa)  # :401-415
    assert set(pl) == {b"MDLQENLVSDAGDDHMV", b"DIVVEPHSNRDIGIVDE", b"FNIGGDVGFSGDLDLEP"}
    rc, pl = ob.parse_pattern_list(raw, lowercase=True)  # :418-431
    assert set(pl) == {b"a" * 32, b"aaaattgcatgaatattgtagatcaaagcaca", b"ctccgaagaagttgctgttcttgatggttatt"}
    # messy file: the whitespace-only survivor is dropped by retain(!is_empty) (helpers.rs:124)
    rc, raw_m = _kmers(golden, "kmers-messy.txt")
    rc, pl = ob.parse_pattern_list(raw_m)
    assert pl == [b"AAAAAAAAAAAAAAAAAAAAAAAAAAAA", b"CTCCGAAGAAGTTGCTGTTCTTGATGGTTATT", b"TTGCATGAATATTGTA"]


def test_reverse_complement_iupac():
    assert ob.reverse_complement(b"ACGT") == b"ACGT"
    assert ob.reverse_complement(b"AACG") == b"CGTT"
    assert ob.reverse_complement(b"acgtn") == b"nacgt"
    assert ob.reverse_complement(b"RYKMBVDHSWN") == b"NWSDHBVKMRY"
    assert ob.reverse_complement(b"A-U*x") == b"x*U-T"  # pass-through bytes
    assert ob.canonical(b"TTTT") == b"AAAA" and ob.canonical(b"AAAC") == b"AAAC"


# ------------------------------------------------------------------ fixture helpers
def _json_rows(j):
    return [(m["file"], m["record_id"], m["pattern"], int(m["position"])) for m in j["matching_records"]]


def _named_rows(rows, files, ids, patterns):
    return [(files[f], ids[f][r].decode(), patterns[p].decode(), pos) for (f, r, p, pos) in rows]


def _text_log_rows(path):
    rows = []
    for ln in open(path).read().split("\n"):
        if ln and not ln.startswith("#"):
            f, r, p, pos = ln.split("\t")
            rows.append((f, r, p, int(pos)))
    return rows


def _check_summary(j, patterns, c):
    s = j["summary_statistics"]
    assert s["number_of_patterns_searched"] == len(patterns)
    assert s["number_of_records_searched"] == c["records"]
    assert s["number_of_characters_searched"] == c["bases"]
    assert s["number_of_matches"] == sum(c["hits"])
    assert s["number_of_distinct_records_with_a_hit"] == sum(c["records_hit"])
    assert s["number_of_patterns_found"] == sum(1 for x in c["pattern_hit_counts"] if x)
    assert j["pattern_hit_counts"] == {p.decode(): n for p, n in zip(patterns, c["pattern_hit_counts"])}


def _matcher_for(raw, rc_flag=False, q=None, force_ac=False, ci=False):
    rc, patterns = ob.parse_pattern_list(raw, reverse_complement=rc_flag)
    assert rc == 0
    use_ac = ob.select_aho_corasick(ci, force_ac, q is not None, patterns)
    return patterns, ob.Matcher(patterns, use_ac, q or 0, ci)


# ------------------------------------------------------------------ extract fixtures
@pytest.mark.parametrize("name,invert", [("simple", False), ("simple-inv", True)])
def test_extract_simple_fixture(golden, name, invert):
    # cmd_extract.rs:886-964: extract -i simple.fasta -r -s ACG [-v]
    recs = textio.read_fastx(os.path.join(golden, "fixtures/input/simple.fasta"))
    patterns, m = _matcher_for([b"ACG"], rc_flag=True)
    assert patterns == [b"ACG", b"CGT"] and not m.use_ac
    keep, rows, c = ob.extract_single(m, [s for _, s in recs], logging=True, invert=invert)
    j = json.load(open(os.path.join(golden, f"fixtures/extract/{name}.json")))
    named = _named_rows(rows, ["simple.fasta"], [[i for i, _ in recs]], patterns)
    assert named == _json_rows(j) == _text_log_rows(os.path.join(golden, f"fixtures/extract/{name}.log"))
    _check_summary(j, patterns, c)
    assert j["meta_information"]["search_algorithm"] == "BNDMq"
    assert j["paired_end_reads_statistics"]["number_of_extracted_records"] == c["extracted"]
    out = textio.read_fastx(os.path.join(golden, f"fixtures/extract/{name}.extracted.fasta"))
    assert [r for r, k in zip(recs, keep) if k] == out
    # logging off must keep the same records
    keep2, _, _ = ob.extract_single(m, [s for _, s in recs], logging=False, invert=invert)
    assert keep2 == keep


def test_extract_fixed_width_fixture(golden):
    # cmd_extract.rs:969-1007: protein pattern, hit spanning a FASTA line break at 79
    recs = textio.read_fastx(os.path.join(golden, "fixtures/input/fixed-width.faa"))
    patterns, m = _matcher_for([b"DKAT"])
    keep, rows, c = ob.extract_single(m, [s for _, s in recs])
    j = json.load(open(os.path.join(golden, "fixtures/extract/fixed-width.json")))
    assert _named_rows(rows, ["fixed-width.faa"], [[i for i, _ in recs]], patterns) == _json_rows(j)
    assert [r[3] for r in rows] == [79, 272]
    _check_summary(j, patterns, c)
    assert c["bases"] == 280 and keep == [True]


def test_extract_paired_fixture(golden):
    # cmd_extract.rs:1012-1056: extract -i paired-1.fastq -2 paired-2.fastq -s CTT
    r1 = textio.read_fastx(os.path.join(golden, "fixtures/input/paired-1.fastq"))
    r2 = textio.read_fastx(os.path.join(golden, "fixtures/input/paired-2.fastq"))
    patterns, m = _matcher_for([b"CTT"])
    keep, rows, c = ob.extract_paired(m, [s for _, s in r1], [s for _, s in r2])
    j = json.load(open(os.path.join(golden, "fixtures/extract/paired.json")))
    named = _named_rows(rows, ["paired-1.fastq", "paired-2.fastq"], [[i for i, _ in r1], [i for i, _ in r2]], patterns)
    assert named == _json_rows(j) == _text_log_rows(os.path.join(golden, "fixtures/extract/paired.log"))
    _check_summary(j, patterns, c)
    p = j["paired_end_reads_statistics"]
    assert (p["number_of_hits_in_file_1"], p["number_of_hits_in_file_2"]) == c["hits"]
    assert (p["number_of_distinct_records_with_a_hit_in_file_1"],
            p["number_of_distinct_records_with_a_hit_in_file_2"]) == c["records_hit"]
    assert p["number_of_extracted_records"] == c["extracted"] == 4
    assert keep == [True, True]
    assert c["pattern_hit_counts"] == [2]  # one pattern, counted once per mate (cmd_extract.rs:575-584)
    assert ob.extract_paired(m, [b"A"], [b"A", b"C"])[0] == ob.MKO_E_PAIR_MISMATCH


# ------------------------------------------------------------------ tag fixtures
@pytest.mark.parametrize("name,flt,inv,out", [
    ("simple", True, False, "simple.extracted.sam"),
    ("simple-inv", False, True, "simple-inv.extracted.sam"),
    ("simple-bam", False, False, "simple.tagged.extracted.sam"),
])
def test_tag_fixtures(golden, name, flt, inv, out):
    # cmd_tag.rs:1011-1132: tag -s CTC -r [-m|-v]; BAM input decodes to the same sequences
    hdr, recs = textio.read_sam(os.path.join(golden, "fixtures/input/simple.sam"))
    patterns, m = _matcher_for([b"CTC"], rc_flag=True)
    assert patterns == [b"CTC", b"GAG"]
    keep, rows, c, found = ob.tag_records(m, [r[9] for r in recs], logging=True, filter_matching=flt, invert=inv)
    j = json.load(open(os.path.join(golden, f"fixtures/tag/{name}.json")))
    fname = j["meta_information"]["input_files"]["record_file_1"]
    assert _named_rows(rows, [fname], [[r[0] for r in recs]], patterns) == _json_rows(j) \
        == _text_log_rows(os.path.join(golden, f"fixtures/tag/{name}.log"))
    _check_summary(j, patterns, c)
    _, exp = textio.read_sam(os.path.join(golden, f"fixtures/tag/{out}"))
    got = [r + [b"km:Z:" + ob.tag_value(patterns, f)] for r, k, f in zip(recs, keep, found) if k]
    assert got == exp
    # logging off (find_match per pattern) yields the same tag sets
    keep2, _, _, found2 = ob.tag_records(m, [r[9] for r in recs], logging=False, filter_matching=flt, invert=inv)
    assert keep2 == keep and [sorted(set(f)) for f in found2] == [sorted(set(f)) for f in found]


def test_tag_value_merge():
    pats = [b"AAC", b"CTC", b"GAG"]
    assert ob.tag_value(pats, []) == b""
    assert ob.tag_value(pats, [2, 1, 1]) == b"CTC,GAG"
    assert ob.tag_value(pats, [2], b"ZZZ,AAC") == b"AAC,GAG,ZZZ"  # cmd_tag.rs:470-490
    assert ob.tag_value(pats, [2], b"") == b"GAG"


# ------------------------------------------------------------------ the AC known-answer vector
def test_aho_corasick_log_json(golden):
    """tests/fixtures/extract/log.json: tag -i simple.bam -S -s CTC AC CT AA T A C G GA AG -r -j
    -> 14 patterns => Aho-Corasick; the only AC known-answer vector the reference holds."""
    j = json.load(open(os.path.join(golden, "fixtures/extract/log.json")))
    hdr, recs = textio.read_sam(os.path.join(golden, "fixtures/input/simple.sam"))
    raw = [b"CTC", b"AC", b"CT", b"AA", b"T", b"A", b"C", b"G", b"GA", b"AG"]
    patterns, m = _matcher_for(raw, rc_flag=True)
    assert len(patterns) == 14 and m.use_ac
    assert j["meta_information"]["search_algorithm"] == "Aho-Corasick"
    keep, rows, c, found = ob.tag_records(m, [r[9] for r in recs], logging=True)
    assert len(rows) == 96
    assert _named_rows(rows, ["simple.bam"], [[r[0] for r in recs]], patterns) == _json_rows(j)
    _check_summary(j, patterns, c)  # AC counts every hit ("A": 13)
    # the independent naive enumerator agrees on order too
    for ridx, r in enumerate(recs):
        assert [(p, s) for (f, rr, p, s) in rows if rr == ridx] == naive.ac_order(patterns, r[9])


# ------------------------------------------------------------------ example-minimal (BASELINE config 1)
def test_example_minimal(golden):
    recs = textio.read_fastx(os.path.join(golden, "example-minimal/sample.fasta"))
    rc, raw = ob.read_kmers_from_text(open(os.path.join(golden, "example-minimal/kmers.txt"), "rb").read())
    patterns, m = _matcher_for(raw)
    assert patterns == [b"AAC"] and not m.use_ac and ob.tune_q_value(3) == 2
    assert [len(s) for _, s in recs] == [897, 898]
    keep, rows, c = ob.extract_single(m, [s for _, s in recs])
    assert keep == [True, True]
    assert [pos for (f, r, p, pos) in rows if r == 0] == [48, 54, 321, 450, 486, 637, 741, 849]
    assert [pos for (f, r, p, pos) in rows if r == 1] == [83, 109, 399, 451, 556, 592, 642, 752, 788, 832, 851]


# ------------------------------------------------------------------ example-workflow goldens
def test_example_workflow_extract(golden):
    wf = os.path.join(golden, "example-workflow")
    r1 = textio.read_fastx(os.path.join(wf, "data/mutant_R1.subset.fastq.gz"))
    r2 = textio.read_fastx(os.path.join(wf, "data/mutant_R2.subset.fastq.gz"))
    meta = json.load(open(os.path.join(wf, "data/subset.meta.json")))
    rc, raw = ob.read_kmers_from_text(open(os.path.join(wf, "significant_kmers.txt"), "rb").read())
    patterns, m = _matcher_for(raw, rc_flag=True)
    assert len(patterns) == 6 and not m.use_ac
    keep, rows, c = ob.extract_paired(m, [s for _, s in r1], [s for _, s in r2])
    j = json.load(open(os.path.join(wf, "logs/mutant_extracted.stats.json")))
    named = _named_rows(rows, ["mutant_R1.fastq", "mutant_R2.fastq"], [[i for i, _ in r1], [i for i, _ in r2]], patterns)
    assert named == _json_rows(j)  # the 36 rows, in order
    assert j["pattern_hit_counts"] == {p.decode(): n for p, n in zip(patterns, c["pattern_hit_counts"])}
    p = j["paired_end_reads_statistics"]
    assert c["hits"] == (13, 23) == (p["number_of_hits_in_file_1"], p["number_of_hits_in_file_2"])
    assert c["records_hit"] == (9, 15) and c["extracted"] == 48 == p["number_of_extracted_records"]
    assert c["records"] == 2 * meta["pairs_kept"] and c["bases"] == meta["bases_kept"]
    e1 = textio.read_fastx(os.path.join(wf, "output/mutant_extracted_1.fastq"))
    e2 = textio.read_fastx(os.path.join(wf, "output/mutant_extracted_2.fastq"))
    assert [r for r, k in zip(r1, keep) if k] == e1 and [r for r, k in zip(r2, keep) if k] == e2


def test_example_workflow_tag(golden):
    # README.md:259: merkurio tag -i mutant_extracted.sorted.sam -f kmers -r -o ...tagged.sam (no logging)
    wf = os.path.join(golden, "example-workflow")
    hdr, recs = textio.read_sam(os.path.join(wf, "output/mutant_extracted.sorted.sam"))
    ehdr, exp = textio.read_sam(os.path.join(wf, "output/mutant_extracted.sorted.tagged.sam"))
    rc, raw = ob.read_kmers_from_text(open(os.path.join(wf, "significant_kmers.txt"), "rb").read())
    patterns, m = _matcher_for(raw, rc_flag=True)
    keep, rows, c, found = ob.tag_records(m, [r[9] for r in recs], logging=False)
    assert all(keep) and len(recs) == 48
    got = [r + [b"km:Z:" + ob.tag_value(patterns, f)] for r, f in zip(recs, found)]
    assert got == exp
    assert [h for h in ehdr if not h.startswith(b"@PG\tID:merkurio")] == hdr


# ------------------------------------------------------------------ differential: oracle vs naive
@pytest.mark.parametrize("alpha", [b"A", b"AC", b"ACGT", b"ACGTN"])
def test_bndmq_equals_naive_random(alpha):
    rnd = random.Random(1234 + len(alpha))
    for _ in range(1500):
        m = rnd.choice([1, 2, 3, 4, 5, 8, 9, 16, 30, 31, 32, 33, 55, 56, 63, 64])
        pat = bytes(rnd.choice(alpha) for _ in range(m))
        n = rnd.randrange(0, 200)
        text = bytes(rnd.choice(alpha) for _ in range(n))
        if rnd.random() < 0.5 and n >= m:
            k = rnd.randrange(0, n - m + 1)
            text = text[:k] + pat + text[k + m:]
        q = rnd.choice([1, 2, ob.tune_q_value(m), m])
        q = max(1, min(q, m))
        b = ob.BNDMq(pat, q)
        exp = naive.occurrences(pat, text)
        assert b.find_all(text) == exp
        assert b.find_match(text) == bool(exp)
        assert ob.bndm_find_all(pat, text) == exp


@pytest.mark.parametrize("ci", [False, True])
def test_ac_equals_naive_random(ci):
    rnd = random.Random(99 + ci)
    alpha = b"ACGTacgtN"
    for _ in range(300):
        npat = rnd.randrange(1, 30)
        raw = [bytes(rnd.choice(alpha) for _ in range(rnd.choice([1, 2, 3, 5, 8, 21, 31, 70]))) for _ in range(npat)]
        rc, patterns = ob.parse_pattern_list(raw)
        ac = ob.AhoCorasick(patterns, ci)
        text = bytes(rnd.choice(alpha) for _ in range(rnd.randrange(0, 300)))
        for p in rnd.sample(patterns, min(3, len(patterns))):
            k = rnd.randrange(0, len(text) + 1)
            text = text[:k] + p + text[k:]
        exp = naive.ac_order(patterns, text, ci)
        assert ac.find_overlapping(text) == exp
        assert ac.is_match(text) == bool(exp)

"""CPU-side checks of the product library: it loads, exports every symbol include/merkurio_hip.h
declares, its host logic (pattern-list preparation, pure helper functions) agrees with the
oracle and the reference KATs, and it fails loudly without a GPU (no CPU fallback)."""
import os
import random
import re

import pytest

import oracle_binding as ob
from merkurio_amd import native as mk

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "merkurio_hip.h")).read()
    declared = set(re.findall(r"\b(mk_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"mk_matcher", "mk_hit", "mk_row", "mk_counters", "mk_codec", "mk_bgzf_member"}
    assert declared == set(mk.EXPORTS), declared ^ set(mk.EXPORTS)
    L = mk.load()
    for name in sorted(declared):
        assert hasattr(L, name), name
    assert L.mk_abi_version() == 7


def test_no_cpu_fallback_without_gpu():
    if mk.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(mk.MerkurioError) as e:
        mk.Matcher([b"ACG"])
    assert e.value.code == mk.MK_E_HIP and "no CPU fallback" in str(e.value)


def test_matcher_validation_precedes_device_use():
    # PatternError mapping does not need a GPU (src/pattern_matching.rs:61-78)
    with pytest.raises(mk.PatternError) as e:
        mk.Matcher([b"abc"], algo=mk.MK_ALGO_BNDMQ, q=4)
    assert e.value.kind == "InvalidQGramLength"
    with pytest.raises(mk.PatternError) as e:
        mk.Matcher([b"a" * 70], algo=mk.MK_ALGO_BNDMQ, q=3)
    assert e.value.kind == "PatternTooLong"
    with pytest.raises(mk.PatternError) as e:
        mk.Matcher([b"a" * 70], algo=mk.MK_ALGO_BNDMQ)  # tune_q_value bails
    assert e.value.kind == "PatternTooLong"
    with pytest.raises(mk.MerkurioError) as e:
        mk.Matcher([])
    assert e.value.code == mk.MK_E_NO_PATTERNS


def test_pure_helpers_match_reference_kats():
    masks, accept = mk.generate_masks(b"abc")  # src/pattern_preprocessing.rs:54-68
    assert accept == 4 and (masks[97], masks[98], masks[99]) == (4, 2, 1)
    masks, accept = mk.generate_masks(b"3$$X3")
    assert (masks[51], masks[36], masks[88], accept) == (17, 12, 2, 16)
    with pytest.raises(mk.PatternError):
        mk.generate_masks(b"x" * 65)
    assert mk.tune_q_value("AAAAAAAACCCCCCCCGGGGGGGGTTTTTTT") == 5  # src/pattern_matching.rs:485-488
    assert [mk.tune_q_value("x" * n) for n in (1, 2, 3, 4, 8, 9, 30, 31, 55, 56, 64)] == [1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6]
    with pytest.raises(mk.MerkurioError):
        mk.tune_q_value("x" * 65)
    assert mk.recommend_aho_corasick([b"AAA", b"CCC"]) is False  # src/helpers.rs:555-567
    assert mk.recommend_aho_corasick([b"A" * 65]) is True
    assert mk.recommend_aho_corasick([b"A"] * 14) is True


def test_pattern_list_matches_oracle_and_kats(golden):
    d = os.path.join(golden, "data")
    three = [b"AAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAA", b"AAAATTGCATGAATATTGTAGATCAAAGCACA", b"CTCCGAAGAAGTTGCTGTTCTTGATGGTTATT"]
    assert mk.parse_pattern_list(kmer_file=os.path.join(d, "kmers.txt")) == three  # helpers.rs:334-348
    assert sorted(mk.read_kmers_from_file(os.path.join(d, "kmers.fasta"))) == three
    assert len(mk.parse_pattern_list(kmer_file=os.path.join(d, "kmers-duplicates.txt"), reverse_complement=True)) == 4
    assert len(mk.parse_pattern_list(kmer_file=os.path.join(d, "kmers.txt"), reverse_complement=True)) == 6
    assert b"AATAACCATCAAGAACAGCAACTTCTTCGGAG" in mk.parse_pattern_list(kmer_file=os.path.join(d, "kmers.txt"), canonical=True)
    assert mk.parse_pattern_list(kmer_file=os.path.join(d, "kmers-messy.txt")) == \
        [b"AAAAAAAAAAAAAAAAAAAAAAAAAAAA", b"CTCCGAAGAAGTTGCTGTTCTTGATGGTTATT", b"TTGCATGAATATTGTA"]
    for bad in (lambda: mk.parse_pattern_list(kmer_seq=[""]), lambda: mk.read_kmers_from_file(os.path.join(d, "kmers-empty.txt"))):
        with pytest.raises(mk.MerkurioError) as e:
            bad()
        assert e.value.code == mk.MK_E_NO_PATTERNS
    # randomized agreement with the oracle, all flag combinations
    rnd = random.Random(7)
    alpha = b"ACGTacgtNRYKMBVDHSWn-*U"
    for _ in range(300):
        raw = [bytes(rnd.choice(alpha) for _ in range(rnd.randrange(0, 12))) for _ in range(rnd.randrange(1, 12))]
        kw = dict(reverse_complement=rnd.random() < .5, canonical=rnd.random() < .3, lowercase=rnd.random() < .3,
                  uppercase=rnd.random() < .3)
        rc, exp = ob.parse_pattern_list(raw, **kw)
        if rc:
            with pytest.raises(mk.MerkurioError):
                mk.parse_pattern_list(kmer_seq=raw, **kw)
        else:
            assert mk.parse_pattern_list(kmer_seq=raw, **kw) == exp
        s = raw[0]
        assert mk.reverse_complement(s) == ob.reverse_complement(s) and mk.canonical(s) == ob.canonical(s)
    txt = b"AC\n   \n#x\n>y\n  #notcomment\nGT\r\n\nTT"
    assert mk.read_kmers_from_text(txt) == ob.read_kmers_from_text(txt)[1]


def test_options_are_validated_before_device_use():
    # mk_matcher_create_ex: explicit options struct instead of environment variables
    for bad in (dict(force_stride=3), dict(gbloom_log2_blocks=40)):
        with pytest.raises(mk.MerkurioError) as e:
            mk.Matcher([b"ACGTACGTACGTACGTACGTA"] * 1, options=bad)
        assert e.value.code == mk.MK_E_INVALID_ARG
    opt = mk.MatcherOptions()
    opt.struct_size = 2
    with pytest.raises(mk.MerkurioError) as e:
        mk.Matcher([b"ACGT"], options=opt)
    assert e.value.code == mk.MK_E_INVALID_ARG


def test_product_reads_no_environment_variables():
    # tuning hooks live in mk_matcher_options; a stray variable must not change geometry
    import subprocess
    out = subprocess.run(["strings", mk.lib_path()], capture_output=True, text=True).stdout
    assert "MERKURIO_" not in out


def test_allocation_failure_is_an_error_code_not_an_abort(tmp_path):
    """bad_alloc inside the library comes back as MK_E_NOMEM across the C ABI (a child process with
    a small address-space limit builds a pattern list that needs several GB)."""
    import subprocess
    import sys
    code = f"""
import ctypes as C, resource, sys
import numpy as np
sys.path.insert(0, {ROOT!r})
from merkurio_amd import native as mk
L = mk.load()
n = 6_000_000
data = np.frombuffer(np.random.default_rng(1).integers(65, 69, size=n * 31, dtype=np.uint8).tobytes(), dtype=np.uint8)
off = (np.arange(n + 1, dtype=np.uint64) * 31).astype(np.uint32)
soft = 1_200 << 20   # the vector<string> + copies of 6 M patterns (+ reverse complements) need > 1.2 GB
resource.setrlimit(resource.RLIMIT_AS, (soft, soft))
pb, po, cnt = C.c_void_p(), C.c_void_p(), C.c_uint32()
rc = L.mk_parse_pattern_list(data.ctypes.data, off.ctypes.data, n, 1, 0, 0, 0, C.byref(pb), C.byref(po), C.byref(cnt))
print("rc", rc, L.mk_last_error().decode())
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert f"rc {mk.MK_E_NOMEM} " in r.stdout, r.stdout


def test_geometry_plan_and_length_classes():
    """mk_plan_geometry (host only): the filter geometry / length classes a matcher would get.  The headline set keeps
    its geometry when one 8-mer joins it (the round-3 rule scanned that set at S = 1, q = 8); uniform sets, forced
    strides and sets too large for the LDS filter stay one class; option validation."""
    rnd = random.Random(7)
    head = mk.plan_geometry([31] * 10_000)
    assert head == {"q_gram": 24, "stride": 8, "in_lds": 1, "split_len": 0, "n_short": 0, "q_gram2": 0, "stride2": 0}
    plus8 = mk.plan_geometry([31] * 10_000 + [8])
    assert (plus8["q_gram"], plus8["stride"]) == (24, 8)
    assert (plus8["split_len"], plus8["n_short"], plus8["q_gram2"], plus8["stride2"]) == (31, 1, 5, 4)
    assert mk.plan_geometry([31] * 10_000 + [8], dict(length_classes=1))["stride"] == 1
    assert mk.plan_geometry([31] * 10_000 + [8], dict(force_stride=1))["split_len"] == 0
    assert mk.plan_geometry([31] * 2048)["stride"] == 16 and mk.plan_geometry([21] * 10_000)["q_gram"] == 14
    # the stride is the cost model's choice (samples against candidates: filter false positives + true q-gram matches)
    mixed = mk.plan_geometry([rnd.randrange(15, 32) for _ in range(10_000)])
    assert (mixed["stride"], mixed["q_gram"], mixed["split_len"]) == (4, 12, 0)
    twelve, nine = mk.plan_geometry([12] * 10_000), mk.plan_geometry([9] * 5)
    assert (twelve["stride"], twelve["q_gram"]) == (2, 11) and (nine["stride"], nine["q_gram"]) == (4, 6)
    # never a stride that puts thousands of entries on one q-gram (10 001 patterns at S = 8 would share 4 one-base keys)
    assert mk.plan_geometry([31] * 10_000 + [8], dict(length_classes=1))["q_gram"] == 8
    # main filter in global memory: patterns below 15 bases (no 14-base q-gram at a stride of 2) form the short class
    big = mk.plan_geometry([21] * 500_000 + [8])
    assert (big["in_lds"], big["q_gram"], big["stride"]) == (0, 14, 8) and (big["split_len"], big["n_short"], big["stride2"]) == (21, 1, 4)
    assert mk.plan_geometry([21] * 500_000 + [8], dict(length_classes=1))["stride"] == 1  # (the round-3 geometry)
    assert mk.plan_geometry([21] * 500_000 + [17])["split_len"] == 0  # 17 bases still admit q = 14 at stride 4
    many = mk.plan_geometry([31] * 10_000 + [10] * 100)
    assert many["split_len"] == 31 and many["n_short"] == 100 and many["q_gram2"] >= 7
    # a short class never samples past its shortest pattern: stride + q - 1 <= length
    for lens in ([31] * 50 + [3], [31] * 50 + [1], [40] * 50 + [5, 9], [20] * 500 + [6] * 3):
        g = mk.plan_geometry(lens, dict(length_classes=2))
        assert g["split_len"] == max(lens) and g["stride2"] + g["q_gram2"] - 1 <= min(lens), (lens[-3:], g)
    with pytest.raises(mk.MerkurioError):
        mk.plan_geometry([31, 8], dict(force_stride2=3))
    with pytest.raises(mk.MerkurioError):
        mk.plan_geometry([31, 8], dict(length_classes=3))
    with pytest.raises(mk.MerkurioError):
        mk.plan_geometry([31, 8], dict(force_q2=9))


def test_bgzf_member_walk_and_codec_without_gpu():
    """mk_bgzf_members is host code (the BSIZE chain of SAM spec 4.1): members written by zlib, extra subfields in front of
    BC, a trailing partial member, headers that are not BGZF.  The codec itself has no CPU path: no device, no handle."""
    import struct
    import zlib

    def member(block, extra=b""):
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        z = co.compress(block) + co.flush()
        xtra = extra + b"BC" + struct.pack("<HH", 2, len(z) + 12 + len(extra) + 6 + 8 - 1)
        return bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff]) + struct.pack("<H", len(xtra)) + xtra + z + struct.pack("<II", zlib.crc32(block), len(block))

    rnd = random.Random(5)
    blocks = [bytes(rnd.choice(b"ACGT") for _ in range(n)) for n in (1, 700, 65280, 0, 3000)]
    blob = b"".join(member(b, b"XY\x03\x00abc" if k == 2 else b"") for k, b in enumerate(blocks)) + mk.bgzf_eof()
    mem, used, text = mk.bgzf_members(blob)
    assert used == len(blob) and text == sum(map(len, blocks)) and len(mem) == len(blocks) + 1
    at = 0
    for m, b in zip(mem, blocks + [b""]):
        assert int(m["isize"]) == len(b) and int(m["out_off"]) == at and int(m["crc"]) == zlib.crc32(b)
        d = zlib.decompressobj(-15)
        assert d.decompress(blob[int(m["data_off"]):int(m["data_off"]) + int(m["data_len"])]) == b
        at += len(b)
    mem2, used2, _ = mk.bgzf_members(blob[:-40])  # the last members are cut: whole members only, no error
    assert used2 < len(blob) - 40 and len(mem2) == len(mem) - 2
    assert mk.bgzf_eof() == bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    for bad in (b"\x1f\x8b\x08\x00" + bytes(30), b"PK\x03\x04" + bytes(30), blob[:3] + b"\x00" + blob[4:]):
        with pytest.raises(mk.MerkurioError) as e:
            mk.bgzf_members(bad)
        assert e.value.code == mk.MK_E_CORRUPT
    L = mk.load()
    assert L.mk_bgzf_deflate_bound(0, 0) == 0 and L.mk_bgzf_deflate_bound(65280, 0) == 65280 + 31 and L.mk_bgzf_deflate_bound(65281, 0) == 65281 + 62
    if mk.device_count() == 0:
        with pytest.raises(mk.MerkurioError) as e:
            mk.Codec()
        assert e.value.code == mk.MK_E_HIP and "no CPU path" in str(e.value)

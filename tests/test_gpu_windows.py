"""Text windows of one or two input files on the device (mk_extract_window, ABI v6) against the oracle's loop
bodies: FASTA (header lines, wrapped sequence lines reach the matcher without their line ends -- a hit may span a line break,
tests/fixtures/extract/fixed-width.log:8), paired FASTQ (src/cmd_extract.rs:463-612: record i of file 1 with record i of file 2,
a pair kept if either mate hits), windows that end anywhere with the unfinished text carried as the next head, plain text and
BGZF members, and the refusals the caller's own reader then takes.  The reference's own paired and FASTA fixtures run through it."""
import os
import random
import zlib

import numpy as np
import pytest

import oracle_binding as ob

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def mk():
    from merkurio_amd import native
    native.load()
    if native.device_count() < 1:
        pytest.fail("no HIP device visible: the gpu-marked tests need an MI355X")
    return native


def _rand(rnd, n, alpha=b"ACGT"):
    return bytes(rnd.choice(alpha) for _ in range(n))


def _fastq(recs, eol=b"\n", final_eol=True):
    text = b"".join(b"@" + rid + eol + seq + eol + b"+" + eol + qual + eol for rid, seq, qual in recs)
    return text if final_eol else text[:-len(eol)]


def _reads(seed, n, lens, patterns, tag=b""):
    rnd = random.Random(seed)
    recs = []
    for i in range(n):
        L = rnd.choice(lens)
        s = bytearray(_rand(rnd, L))
        if patterns and L >= 31 and rnd.random() < 0.15:
            p = rnd.choice(patterns)
            k = rnd.randrange(0, L - len(p) + 1)
            s[k:k + len(p)] = p
        q = _rand(rnd, L, b"@+IJ#5ACGT>")
        recs.append((b"r%d%s extra" % (i, tag), bytes(s), q))
    return recs


def _bgzf(data, block, level=6):
    import struct
    out = bytearray()
    for b in range(0, len(data), block):
        chunk = data[b:b + block]
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        payload = co.compress(chunk) + co.flush()
        out += bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0]) + struct.pack("<H", 18 + len(payload) + 8 - 1) + payload
        out += struct.pack("<II", zlib.crc32(chunk), len(chunk))
    return bytes(out)


# ---- FASTA -------------------------------------------------------------------------------------------------------------------
def _fasta(recs, width, eol=b"\n", final_eol=True):
    """records (id, sequence) wrapped at `width` columns (0 = one line)"""
    out = []
    for rid, seq in recs:
        out.append(b">" + rid + eol)
        if width:
            for k in range(0, len(seq), width):
                out.append(seq[k:k + width] + eol)
        else:
            out.append(seq + eol)
    text = b"".join(out)
    return text if final_eol else text[:-len(eol)]


def _parse_fasta(text):
    """what needletail makes of it (cli/io.cpp restates the same): record starts, ids, sequences without '\\n' / '\\r'"""
    starts, ids, seqs = [], [], []
    pos = 0
    for line in text.split(b"\n"):
        if line.startswith(b">"):
            starts.append(pos)
            ids.append(line[1:].rstrip(b"\r"))
            seqs.append(bytearray())
        elif seqs:
            seqs[-1] += line.replace(b"\r", b"")
        pos += len(line) + 1
    return starts, ids, [bytes(s) for s in seqs]


@pytest.mark.parametrize("width,eol,final_eol", [(60, b"\n", True), (0, b"\n", True), (70, b"\r\n", True), (61, b"\n", False), (1, b"\n", True)])
def test_fasta_window_equals_oracle(mk, width, eol, final_eol):
    rnd = random.Random(width * 7 + len(eol))
    patterns = mk.parse_pattern_list(kmer_seq=[_rand(rnd, 21) for _ in range(200)], reverse_complement=True)
    recs = []
    for i in range(3000 if width != 1 else 300):
        L = rnd.choice([0, 1, 20, 21, 59, 60, 61, 120, 500, 3000])
        s = bytearray(_rand(rnd, L, b"ACGTNacgt"))
        if L >= 21 and rnd.random() < 0.3:  # planted so that occurrences cross line breaks at every phase
            p = rnd.choice(patterns)
            k = rnd.randrange(0, L - 21 + 1)
            s[k:k + 21] = p
        recs.append((b"chr%d len=%d" % (i, L), bytes(s)))
    text = _fasta(recs, width, eol, final_eol)
    starts, ids, seqs = _parse_fasta(text)
    assert seqs == [s for _, s in recs]
    m = mk.Matcher(patterns)
    om = ob.Matcher(patterns, True, 0, False)
    for logging in (True, False):
        for invert in (False, True):
            r = m.extract_window([{"text": text}], fmt=mk.MK_TEXT_FASTA, logging=logging, invert=invert, want=("tail", "kept"))
            k_o, r_o, c_o = ob.extract_single(om, seqs, logging=logging, invert=invert)
            assert r["status"] == 0 and r["n_rec"] == len(recs)
            assert r["keep"] == k_o and r["rows"] == r_o and r["counters"] == c_o
            S = r["sources"][0]
            assert S["rec_start"] == starts + [len(text)] and S["n_used"] == len(text) and S["tail"] == b""
            ends = starts[1:] + [len(text)]
            assert S["kept"] == b"".join(text[a:b] for a, b, k in zip(starts, ends, k_o) if k)
    assert any(k_o) and c_o["extracted"] < len(recs)


def test_fasta_window_that_ends_anywhere_leaves_its_last_record_as_the_tail(mk):
    rnd = random.Random(11)
    patterns = mk.parse_pattern_list(kmer_seq=[_rand(rnd, 15) for _ in range(50)])
    recs = [(b"s%d" % i, _rand(rnd, rnd.randrange(10, 400)) + patterns[i % 50]) for i in range(400)]
    text = _fasta(recs, 80)
    starts, ids, seqs = _parse_fasta(text)
    m = mk.Matcher(patterns)
    om = ob.Matcher(patterns, True, 0, False)
    # cut in the middle of record 250's sequence, in its header, and right behind a record's last line end
    for cut in (starts[250] + 40, starts[250] + 2, starts[250]):
        r = m.extract_window([{"text": text[:cut], "ends_at_record": False}], fmt=mk.MK_TEXT_FASTA)
        n = 250 if cut > starts[250] else 249  # (cut at a record start: the record in front of it cannot be known to be whole)
        assert r["status"] == 0 and r["n_rec"] == n and r["sources"][0]["n_used"] == starts[n] and r["sources"][0]["tail"] == text[starts[n]:cut]
        k_o, r_o, c_o = ob.extract_single(om, seqs[:n])
        assert r["keep"] == k_o and r["rows"] == r_o and r["counters"] == c_o
        # ... and the rest of the file with that tail as its head
        r2 = m.extract_window([{"head": r["sources"][0]["tail"], "text": text[cut:]}], fmt=mk.MK_TEXT_FASTA)
        k2, rows2, c2 = ob.extract_single(om, seqs[n:])
        assert r2["n_rec"] == len(recs) - n and r2["keep"] == k2 and r2["rows"] == rows2 and r2["counters"] == c2
    # text in front of the first header, an empty window
    assert m.extract_window([{"text": b"\n" + text}], fmt=mk.MK_TEXT_FASTA)["status"] == 1
    assert m.extract_window([{"text": b"ACGT\n" + text}], fmt=mk.MK_TEXT_FASTA)["status"] == 1
    assert m.extract_window([{"text": b""}], fmt=mk.MK_TEXT_FASTA)["n_rec"] == 0


def test_fasta_chromosome_sized_records(mk):
    """few, long records: one unwrapped 40 Mbp line, one wrapped 25 Mbp record, small ones around them; the kept records come back
    packed (records above 1 MiB are copied one by one)"""
    rnd = np.random.default_rng(5)
    kmers = [b"ACGTTGCAAGGCTTAAGGCCATTGACCA", b"TTGACCAGGTACCATTTGGACCAAGGTT"]
    patterns = mk.parse_pattern_list(kmer_seq=kmers)
    big1 = np.frombuffer(b"ACGT", dtype=np.uint8)[rnd.integers(0, 4, 40_000_000)].tobytes()
    big2 = bytearray(np.frombuffer(b"ACGT", dtype=np.uint8)[rnd.integers(0, 4, 25_000_000)].tobytes())
    for k in (5, 59, 60 * 1000 - 10, 24_999_000):  # across line breaks of the 60-column wrapping
        big2[k:k + 28] = kmers[0]
    recs = [(b"small0", b"ACGT" * 10), (b"chrA", big1[:1000] + kmers[1] + big1[1000:]), (b"small1", b"TTTT"), (b"chrB", bytes(big2)),
            (b"small2", kmers[0])]
    text = b"".join([_fasta(recs[:1], 60), _fasta(recs[1:2], 0), _fasta(recs[2:3], 60), _fasta(recs[3:4], 60), _fasta(recs[4:], 60)])
    starts, ids, seqs = _parse_fasta(text)
    m = mk.Matcher(patterns)
    om = ob.Matcher(patterns, m.use_ac, 0, False)  # (two patterns: the reference's rule selects BNDMq)
    assert not m.use_ac
    r = m.extract_window([{"text": text}], fmt=mk.MK_TEXT_FASTA, want=("kept",))
    k_o, r_o, c_o = ob.extract_single(om, seqs)
    assert r["status"] == 0 and r["keep"] == k_o == [False, True, False, True, True] and r["rows"] == r_o and r["counters"] == c_o
    ends = starts[1:] + [len(text)]
    assert r["sources"][0]["kept"] == b"".join(text[a:b] for a, b, k in zip(starts, ends, k_o) if k)


def test_reference_fasta_fixtures_through_the_window_path(mk):
    """tests/fixtures/input/simple.fasta (-r -s ACG) and fixed-width.faa (protein DKAT, a hit across a line break at 79):
    the rows of the reference's own logs (src/cmd_extract.rs:886-1007)"""
    fx = os.path.join(GOLDEN, "fixtures")
    for name, kmers, rc in (("simple.fasta", [b"ACG"], True), ("fixed-width.faa", [b"DKAT"], False)):
        text = open(os.path.join(fx, "input", name), "rb").read()
        patterns = mk.parse_pattern_list(kmer_seq=kmers, reverse_complement=rc)
        m = mk.Matcher(patterns)
        r = m.extract_window([{"text": text}], fmt=mk.MK_TEXT_FASTA)
        starts, ids, seqs = _parse_fasta(text)
        log = open(os.path.join(fx, "extract", name.rsplit(".", 1)[0] + ".log"), "rb").read().split(b"\n")
        want = [tuple(ln.split(b"\t")) for ln in log if ln and not ln.startswith(b"#")]
        got = [(name.encode(), ids[rec], patterns[pat], b"%d" % pos) for _, rec, pat, pos in r["rows"]]
        assert got == want and r["status"] == 0
    assert (b"fixed-width.faa", b"protein1", b"DKAT", b"79") in got


# ---- paired FASTQ --------------------------------------------------------------------------------------------------------------
def _check_pairs(mk, m, om, r, recs1, recs2, n, logging, invert):
    k_o, r_o, c_o = ob.extract_paired(om, [s for _, s, _ in recs1[:n]], [s for _, s, _ in recs2[:n]], logging=logging, invert=invert)
    assert r["status"] == 0 and r["n_rec"] == n
    assert r["keep"] == k_o and r["rows"] == r_o and r["counters"] == c_o
    return k_o


@pytest.mark.parametrize("algo_q", [None, 4])
@pytest.mark.parametrize("lens,eol", [([150], b"\n"), ([36, 75, 150, 151], b"\r\n")])
def test_paired_windows_equal_oracle(mk, lens, eol, algo_q):
    rnd = random.Random(len(lens) + len(eol))
    n_pat = 300 if algo_q is None else 5
    patterns = mk.parse_pattern_list(kmer_seq=[_rand(rnd, 31) for _ in range(n_pat)], reverse_complement=algo_q is None)
    r1, r2 = _reads(1, 6000, lens, patterns, b"/1"), _reads(2, 6000, lens, patterns, b"/2")
    t1, t2 = _fastq(r1, eol), _fastq(r2, eol, final_eol=False)
    m = mk.Matcher(patterns, algo=mk.MK_ALGO_BNDMQ, q=algo_q) if algo_q else mk.Matcher(patterns)
    om = ob.Matcher(patterns, algo_q is None, algo_q or 0, False)
    for logging in (True, False):
        for invert in (False, True):
            r = m.extract_window([{"text": t1}, {"text": t2}], logging=logging, invert=invert, want=("tail", "kept"))
            k_o = _check_pairs(mk, m, om, r, r1, r2, 6000, logging, invert)
            for S, recs, text in ((r["sources"][0], r1, t1), (r["sources"][1], r2, t2)):
                assert S["n_used"] == len(text) and S["tail"] == b"" and S["rec_start"][-1] == len(text)
                a = S["rec_start"]
                assert S["kept"] == b"".join(text[a[i]:a[i + 1]] for i in range(6000) if k_o[i])
    assert any(k_o) and not all(k_o)


def test_paired_windows_of_unequal_record_counts_carry_their_leftovers(mk):
    """the two files' windows hold different numbers of records (ids and reads of different lengths): a call pairs what both have,
    the rest is the tail; chained over a whole pair of files == the oracle on all pairs; a file that ends early leaves the other's
    records unpaired (src/cmd_extract.rs:465-468,608-612: the caller words that error)"""
    rnd = random.Random(5)
    patterns = mk.parse_pattern_list(kmer_seq=[_rand(rnd, 31) for _ in range(100)], reverse_complement=True)
    r1 = _reads(3, 5000, [100, 150], patterns, b"/1 a much longer header line than the mate's")
    r2 = _reads(4, 5000, [60, 150, 250], patterns, b"/2")
    t1, t2 = _fastq(r1), _fastq(r2)
    m = mk.Matcher(patterns)
    om = ob.Matcher(patterns, True, 0, False)
    k_all, rows_all, c_all = ob.extract_paired(om, [s for _, s, _ in r1], [s for _, s, _ in r2], logging=True)
    W = 200_000
    pos, heads = [0, 0], [b"", b""]
    keep, rows, base = [], [], 0
    cnt = None
    rounds = 0
    while pos[0] < len(t1) or pos[1] < len(t2) or heads[0] or heads[1]:
        src = []
        for k, t in enumerate((t1, t2)):
            body = t[pos[k]:pos[k] + W - len(heads[k])] if len(heads[k]) < W else b""
            pos[k] += len(body)
            src.append({"head": heads[k], "text": body, "ends_at_record": pos[k] >= len(t)})
        r = m.extract_window(src, logging=True)
        assert r["status"] == 0
        heads = [r["sources"][k]["tail"] for k in range(2)]
        keep += r["keep"]
        rows += [(f, rec + base, p, o) for f, rec, p, o in r["rows"]]
        base += r["n_rec"]
        c = r["counters"]
        if cnt is None:
            cnt = c
        else:
            for key in ("records", "bases", "extracted"):
                cnt[key] += c[key]
            for key in ("hits", "records_hit"):
                cnt[key] = tuple(a + b for a, b in zip(cnt[key], c[key]))
            cnt["pattern_hit_counts"] = [a + b for a, b in zip(cnt["pattern_hit_counts"], c["pattern_hit_counts"])]
        rounds += 1
        assert rounds < 100
    assert rounds > 5 and keep == k_all and rows == rows_all and cnt == c_all
    # file 2 three records short: the window pairs what there is; file 1's last three records stay behind as its tail
    short = _fastq(r2[:4997])
    r = m.extract_window([{"text": t1}, {"text": short}], logging=False)
    assert r["n_rec"] == 4997 and r["sources"][0]["n_rec_seen"] == 5000 and r["sources"][1]["n_rec_seen"] == 4997
    assert r["sources"][0]["tail"] == _fastq(r1[4997:]) and r["sources"][1]["tail"] == b""
    assert r["keep"] == k_all[:4997]
    # ... and nothing at all in one of them
    r = m.extract_window([{"text": t1[:3000], "ends_at_record": False}, {"text": b""}])
    assert r["n_rec"] == 0 and r["sources"][0]["n_used"] == 0 and r["sources"][0]["tail"] == t1[:3000]


def test_paired_bgzf_members_with_heads(mk):
    """both mates bgzip'ed (zlib level-6 members of 3000 ... 65280 bytes), windows of a few members with the unfinished records
    carried; the kept records of both files come back packed"""
    rnd = random.Random(9)
    patterns = mk.parse_pattern_list(kmer_seq=[_rand(rnd, 31) for _ in range(100)], reverse_complement=True)
    r1, r2 = _reads(5, 4000, [150], patterns, b"/1"), _reads(6, 4000, [75, 150], patterns, b"/2")
    t1, t2 = _fastq(r1), _fastq(r2)
    blobs = [_bgzf(t1, 20_000), _bgzf(t2, 65_280)]
    tabs = [mk.bgzf_members(b)[0] for b in blobs]
    codec = mk.Codec()
    m = mk.Matcher(patterns)
    om = ob.Matcher(patterns, True, 0, False)
    for logging, invert in ((True, False), (False, True)):
        k_all, rows_all, c_all = ob.extract_paired(om, [s for _, s, _ in r1], [s for _, s, _ in r2], logging=logging, invert=invert)
        nxt, heads = [0, 0], [b"", b""]
        keep, rows, base, kept_text = [], [], 0, [b"", b""]
        per = (7, 3)
        while nxt[0] < len(tabs[0]) or nxt[1] < len(tabs[1]) or heads[0] or heads[1]:
            src = []
            for k in range(2):
                grp = tabs[k][nxt[k]:nxt[k] + per[k]]
                nxt[k] += len(grp)
                src.append({"head": heads[k], "blob": blobs[k], "members": grp, "ends_at_record": nxt[k] >= len(tabs[k])})
            r = m.extract_window(src, logging=logging, invert=invert, codec=codec, want=("tail", "kept"))
            assert r["status"] == 0
            heads = [r["sources"][k]["tail"] for k in range(2)]
            keep += r["keep"]
            rows += [(f, rec + base, p, o) for f, rec, p, o in r["rows"]]
            base += r["n_rec"]
            for k in range(2):
                kept_text[k] += r["sources"][k]["kept"]
        assert keep == k_all and rows == rows_all and base == 4000
        for k, recs in enumerate((r1, r2)):
            assert kept_text[k] == b"".join(_fastq([recs[i]]) for i in range(4000) if k_all[i])
    # a damaged member in the second file
    bad = bytearray(blobs[1])
    bad[int(tabs[1][1]["data_off"]) + 40] ^= 0x55
    with pytest.raises(mk.MerkurioError) as e:
        m.extract_window([{"blob": blobs[0], "members": tabs[0]}, {"blob": bytes(bad), "members": tabs[1]}], codec=codec)
    assert e.value.code == mk.MK_E_CORRUPT
    codec.close()


def test_reference_paired_fixture_through_the_window_path(mk):
    """tests/fixtures/input/paired-{1,2}.fastq, -s CTT (src/cmd_extract.rs:1012-1056): rows and the paired summary of paired.log"""
    fx = os.path.join(GOLDEN, "fixtures")
    t1, t2 = (open(os.path.join(fx, f"input/paired-{k}.fastq"), "rb").read() for k in (1, 2))
    patterns = mk.parse_pattern_list(kmer_seq=[b"CTT"])
    m = mk.Matcher(patterns)
    r = m.extract_window([{"text": t1}, {"text": t2}], want=("kept",))
    ids = [[ln[1:] for ln in t.split(b"\n")[0::4] if ln] for t in (t1, t2)]
    got = [(b"paired-%d.fastq" % (f + 1), ids[f][rec], patterns[pat], b"%d" % pos) for f, rec, pat, pos in r["rows"]]
    log = open(os.path.join(fx, "extract/paired.log"), "rb").read().split(b"\n")
    assert got == [tuple(ln.split(b"\t")) for ln in log if ln and not ln.startswith(b"#")]
    c = r["counters"]
    assert c["records"] == 4 and c["bases"] == 32 and c["hits"] == (1, 1) and c["records_hit"] == (1, 1) and c["extracted"] == 4
    # the kept records are what the reference writes (it re-emits 4 lines and adds the final line end: 56 bytes against 55)
    for k in (1, 2):
        want = open(os.path.join(fx, f"extract/paired_{k}.extracted.fastq"), "rb").read()
        kept = r["sources"][k - 1]["kept"]
        assert kept == want[:len(kept)] and len(want) - len(kept) == 1 and want.endswith(b"\n")


def test_window_refusals(mk):
    rnd = random.Random(2)
    patterns = mk.parse_pattern_list(kmer_seq=[_rand(rnd, 31) for _ in range(20)])
    good = _fastq(_reads(7, 50, [150], patterns))
    m = mk.Matcher(patterns)
    lines = good.split(b"\n")
    blank = b"\n".join(lines[:8] + [b""] + lines[8:])
    wrapped = b"\n".join(lines[:1] + [lines[1][:70], lines[1][70:]] + lines[2:])
    for t in (blank, wrapped, good[:-40], good.replace(b"\n+\n", b"\n-\n", 1), b">fa\nACGT\n"):
        assert m.extract_window([{"text": t}])["status"] == 1                       # single
        assert m.extract_window([{"text": good}, {"text": t}])["status"] == 1       # as the second mate
    # a window that ends anywhere is not refused for its unfinished record ...
    r = m.extract_window([{"text": good[:-40], "ends_at_record": False}])
    assert r["status"] == 0 and r["n_rec"] == 49
    # ... but a malformed whole record in it still is
    assert m.extract_window([{"text": wrapped[:-40], "ends_at_record": False}])["status"] == 1
    # FASTQ text handed in as FASTA and the other way round
    assert m.extract_window([{"text": good}], fmt=mk.MK_TEXT_FASTA)["status"] == 1

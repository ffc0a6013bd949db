"""`tag` on windows of BAM text that stay on the device (mk_tag_bam_window, ABI v7) against the oracle's
restatement of process_record (src/cmd_tag.rs:387-497): the record chain indexed in pieces whose guessed starts are proved, 4-bit
sequences unpacked for the matcher, keep / drop, the tag value appended to the raw record, the output deflated into BGZF members.
Expected bytes are built here from the oracle's answers and the BAM layout (SAM specification 4.2); zlib inflates the device's
members.  The reference's own simple.bam runs through it and must give the records of its tag fixture."""
import gzip
import os
import random
import struct
import zlib

import numpy as np
import pytest

import oracle_binding as ob

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NIB = b"=ACMGRSVTWYHKDBN"


@pytest.fixture(scope="module")
def mk():
    from merkurio_amd import native
    native.load()
    if native.device_count() < 1:
        pytest.fail("no HIP device visible: the gpu-marked tests need an MI355X")
    return native


def _bgzf(data, block=0xff00, level=6):
    out = bytearray()
    for b in range(0, len(data), block):
        chunk = data[b:b + block]
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        payload = co.compress(chunk) + co.flush()
        out += bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0]) + struct.pack("<H", 18 + len(payload) + 8 - 1) + payload
        out += struct.pack("<II", zlib.crc32(chunk), len(chunk))
    return bytes(out)


def bam_record(name, seq, aux=b"", cigar=(), ref=0, pos=100, flag=0, mapq=60):
    """block_size + one BAM record; seq in the 16-letter alphabet"""
    l = len(seq)
    packed = bytearray((l + 1) // 2)
    for k, ch in enumerate(seq):
        packed[k >> 1] |= NIB.index(ch) << (4 if k % 2 == 0 else 0)
    body = struct.pack("<iiBBHHHiiii", ref, pos, len(name) + 1, mapq, 4680, len(cigar), flag, l, -1, -1, 0)
    body += name + b"\0" + b"".join(struct.pack("<I", c) for c in cigar) + bytes(packed) + bytes([30] * l) + aux
    return struct.pack("<i", len(body)) + body


def decode(rec):
    """(name, ASCII sequence) of block_size + record"""
    l_name, n_cig, l_seq = rec[12], struct.unpack_from("<H", rec, 16)[0], struct.unpack_from("<i", rec, 20)[0]
    p = 36 + l_name + 4 * n_cig
    seq = bytes(NIB[(rec[p + (k >> 1)] >> (4 if k % 2 == 0 else 0)) & 15] for k in range(l_seq))
    return rec[36:36 + l_name - 1], seq


def split_records(text):
    out, p = [], 0
    while p + 4 <= len(text):
        b = struct.unpack_from("<i", text, p)[0]
        if p + 4 + b > len(text):
            break
        out.append(text[p:p + 4 + b])
        p += 4 + b
    return out, p


def make_records(rnd, n, patterns, lens=(150,), hit=0.2, alpha=b"ACGT", aux_kinds=True):
    recs = []
    for i in range(n):
        L = rnd.choice(lens)
        s = bytearray(rnd.choice(alpha) for _ in range(L))
        if patterns and rnd.random() < hit:
            for _ in range(rnd.choice((1, 1, 2, 3))):
                p = rnd.choice(patterns)
                if len(p) <= L:
                    k = rnd.randrange(0, L - len(p) + 1)
                    s[k:k + len(p)] = p
        aux = b""
        if aux_kinds:
            pick = rnd.randrange(6)
            if pick >= 1:
                aux += b"NMC" + bytes([rnd.randrange(9)])
            if pick >= 2:
                aux += b"ASi" + struct.pack("<i", rnd.randrange(1000))
            if pick >= 3:
                aux += b"RGZ" + b"grp%d" % rnd.randrange(4) + b"\0"
            if pick >= 4:
                aux += b"ZBB" + b"s" + struct.pack("<i", 3) + struct.pack("<hhh", 1, -2, 3)
            if pick >= 5:
                aux += b"XAA" + b"q" + b"XHH" + b"0AFF\0" + b"XFf" + struct.pack("<f", 1.5) + b"XSs" + struct.pack("<h", -7)
        cigar = ((L << 4) | 0,) if L else ()
        recs.append(bam_record(b"read%d_%d" % (i, rnd.randrange(10 ** 6)), bytes(s), aux, cigar, pos=rnd.randrange(10 ** 6)))
    return recs


def expected(ob_m, patterns, recs, tag, logging, filter_matching, invert, existing=None):
    """oracle: keep, rows [(name, rec, pat, pos)], counters, output text of the kept records with their tag appended
    (existing: per record, the Z value of a field of that name the record already carries, or None)"""
    dec = [decode(r) for r in recs]
    keep, rows, c, found = ob.tag_records(ob_m, [s for _, s in dec], logging=logging, filter_matching=filter_matching, invert=invert)
    out = bytearray()
    for i, (r, k, f) in enumerate(zip(recs, keep, found)):
        if not k:
            continue
        val = ob.tag_value(patterns, f, existing[i] if existing and existing[i] else None)
        body = r[4:] + tag + b"Z" + val + b"\0"
        out += struct.pack("<i", len(body)) + body
    names = [(dec[rec][0], rec, pat, pos) for (_, rec, pat, pos) in rows]
    return keep, names, c, bytes(out)


def run_windows(mk, m, codec, blob, members, cuts, **kw):
    """the members in windows [cuts[i], cuts[i + 1]); tails chained as heads -> list of results"""
    head, res = b"", []
    for i in range(len(cuts) - 1):
        mem = members[cuts[i]:cuts[i + 1]]
        r = m.tag_bam_window(codec, head, blob, mem, last=(i == len(cuts) - 2), **kw)
        res.append(r)
        if r["status"]:
            break
        head = r["tail"]
    return res


PATS31 = None


def patterns31(mk, n=200, seed=3):
    rnd = random.Random(seed)
    raw = [bytes(rnd.choice(b"ACGT") for _ in range(31)) for _ in range(n)]
    return mk.parse_pattern_list(kmer_seq=raw)


@pytest.mark.parametrize("filter_matching,invert", [(False, False), (True, False), (False, True)])
@pytest.mark.parametrize("logging", [True, False])
def test_window_matches_oracle(mk, filter_matching, invert, logging):
    rnd = random.Random(11)
    pats = patterns31(mk)
    recs = make_records(rnd, 3000, pats, lens=(150,))
    text = b"".join(recs)
    blob = _bgzf(text)
    members, used, tb = mk.bgzf_members(blob)
    assert used == len(blob) and tb == len(text)
    m, codec = mk.Matcher(pats, device=0), mk.Codec(0)
    om = ob.Matcher(pats, True, 0, False)
    keep, rows, c, out = expected(om, pats, recs, b"km", logging, filter_matching, invert)
    r = m.tag_bam_window(codec, b"", blob, members, last=True, logging=logging, filter_matching=filter_matching, invert=invert)
    assert r["status"] == 0 and r["n_rec"] == len(recs) and r["n_used"] == len(text) and r["tail"] == b""
    assert r["n_kept"] == sum(keep) and r["out_text_bytes"] == len(out)
    assert gzip.decompress(r["out"] + mk.bgzf_eof()) == out
    if logging:
        assert r["rows"] == rows
        got = dict(r["counters"])
        assert got.pop("extracted") == sum(keep)  # (the device reports the records it wrote; the reference has no such counter in tag)
        want = dict(c)
        want.pop("extracted")
        assert got == want
    # the members are cut every 65280 bytes of output text
    om_, _, _ = mk.bgzf_members(r["out"])
    assert [int(x) for x in om_["isize"]] == [min(65280, len(out) - k) for k in range(0, len(out), 65280)]
    codec.close()


def test_ragged_lengths_heads_and_small_pieces(mk):
    """records of many lengths (0, odd, long), members that end anywhere, windows of a few members with the tail carried, and
    pieces of 256 bytes: every piece start is a guess in the middle of records that has to be proved"""
    rnd = random.Random(5)
    pats = patterns31(mk, 50)
    recs = make_records(rnd, 1200, pats, lens=(0, 1, 31, 32, 75, 149, 150, 151, 600, 2500), hit=0.4, alpha=b"ACGTN")
    text = b"".join(recs)
    blob = _bgzf(text, block=7001)
    members, _, _ = mk.bgzf_members(blob)
    m, codec = mk.Matcher(pats, device=0), mk.Codec(0)
    om = ob.Matcher(pats, True, 0, False)
    keep, rows, c, out = expected(om, pats, recs, b"XK", True, False, False)
    for piece in (256, 4096, 0):
        cuts = list(range(0, len(members), 9)) + [len(members)]
        res = run_windows(mk, m, codec, blob, members, cuts, tag=b"XK", logging=True, piece_bytes=piece, block_bytes=10000)
        assert all(x["status"] == 0 for x in res)
        assert sum(x["n_rec"] for x in res) == len(recs)
        got = b"".join(gzip.decompress(x["out"] + mk.bgzf_eof()) for x in res)
        assert got == out
        got_rows, base = [], 0
        for x in res:
            got_rows += [(nm, rec + base, pat, pos) for (nm, rec, pat, pos) in x["rows"]]
            base += x["n_rec"]
        assert got_rows == rows
        tot = {k: 0 for k in ("records", "bases")}
        for x in res:
            for k in tot:
                tot[k] += x["counters"][k]
        assert tot["records"] == c["records"] and tot["bases"] == c["bases"]
        assert sum(x["counters"]["hits"][0] for x in res) == c["hits"][0]
        assert np.array_equal(np.sum([x["counters"]["pattern_hit_counts"] for x in res], axis=0), c["pattern_hit_counts"])
    codec.close()


def test_bndmq_counts_and_iupac_letters(mk):
    """fewer than 14 patterns: BNDMq's pattern_hit_counts (one per record and pattern) and its emission order; sequences with the
    whole 16-letter alphabet"""
    rnd = random.Random(9)
    pats = mk.parse_pattern_list(kmer_seq=[b"ACGTACG", b"NNRYK", b"GATTACA", b"TTT"])
    recs = make_records(rnd, 800, pats, lens=(40, 41, 90), hit=0.5, alpha=b"ACGTNRYKMSWBDHV=")
    text = b"".join(recs)
    blob = _bgzf(text)
    members, _, _ = mk.bgzf_members(blob)
    m, codec = mk.Matcher(pats, device=0), mk.Codec(0)
    assert not m.use_ac
    om = ob.Matcher(pats, False, 0, False)
    keep, rows, c, out = expected(om, pats, recs, b"km", True, True, False)
    r = m.tag_bam_window(codec, b"", blob, members, last=True, logging=True, filter_matching=True)
    assert r["status"] == 0 and gzip.decompress(r["out"] + mk.bgzf_eof()) == out and r["rows"] == rows
    assert r["counters"]["pattern_hit_counts"] == c["pattern_hit_counts"] and r["counters"]["hits"] == c["hits"]
    codec.close()


def test_refusals(mk):
    rnd = random.Random(2)
    pats = patterns31(mk, 20)
    m, codec = mk.Matcher(pats, device=0), mk.Codec(0)
    good = make_records(rnd, 50, pats, hit=0.5)

    def run(recs, **kw):
        text = b"".join(recs)
        blob = _bgzf(text)
        members, _, _ = mk.bgzf_members(blob)
        return m.tag_bam_window(codec, b"", blob, members, last=True, **kw)

    # a kept record with a field of the tag's name that is not a string (the reference bails), whose value is not plain ASCII (the
    # reference checks UTF-8 first) or is very long: the host path's
    hit_seq = pats[0] + b"A" * 40
    r = run(good + [bam_record(b"old", hit_seq, b"kmi" + struct.pack("<i", 5))] + good)
    assert r["status"] == 4 and r["out"] == b""
    r = run(good + [bam_record(b"old", hit_seq, b"kmZ" + "AAA,\u00e9".encode() + b"\0")])
    assert r["status"] == 4
    r = run(good + [bam_record(b"old", hit_seq, b"kmZ" + b"ACGT," * 500 + b"\0")])
    assert r["status"] == 4
    # ... but not when that record is dropped (-v drops records with a hit; the reference never looks at a dropped record's tags)
    r = run(good + [bam_record(b"old", hit_seq, b"kmi" + struct.pack("<i", 5))], invert=True)
    assert r["status"] == 0
    # optional fields that do not parse
    r = run(good + [bam_record(b"odd", b"ACGT" * 10, b"XX?" + b"1234")])
    assert r["status"] == 2
    r = run(good + [bam_record(b"odd", b"ACGT" * 10, b"XXZ" + b"no terminator")])
    assert r["status"] == 2
    # a block_size below the fixed fields / sizes that do not add up: the serial parser's "truncated file"
    bad = bytearray(good[3])
    struct.pack_into("<i", bad, 0, 20)
    r = run(good[:3] + [bytes(bad)] + good[4:])
    assert r["status"] == 1
    bad = bytearray(good[3])
    struct.pack_into("<i", bad, 20, 10 ** 6)  # l_seq
    r = run(good[:3] + [bytes(bad)] + good[4:])
    assert r["status"] == 1
    # the file ends inside a record
    text = b"".join(good)
    blob = _bgzf(text[:-7])
    members, _, _ = mk.bgzf_members(blob)
    r = m.tag_bam_window(codec, b"", blob, members, last=True)
    assert r["status"] == 8
    r = m.tag_bam_window(codec, b"", blob, members, last=False)  # (more text may follow: the unfinished record is the tail)
    assert r["status"] == 0 and r["n_rec"] == len(good) - 1 and r["tail"] == good[-1][:-7]
    # a damaged member
    blob = bytearray(_bgzf(text))
    blob[len(blob) // 2] ^= 0x55
    members, _, _ = mk.bgzf_members(bytes(blob))
    with pytest.raises(mk.MerkurioError) as e:
        m.tag_bam_window(codec, b"", bytes(blob), members, last=True)
    assert e.value.code == mk.MK_E_CORRUPT
    codec.close()


def test_existing_tag_values_are_merged(mk):
    """records that already carry the tag (src/cmd_tag.rs:470-485): the new field's value = the found patterns and the ','-separated
    items of the FIRST field of that name, sort_unstable + dedup, joined -- unsorted values, duplicates, empty items, items equal to
    found patterns, an empty value ("do nothing"), a second field of the same name (ignored); the old field stays in the record"""
    rnd = random.Random(8)
    pats = patterns31(mk, 30)
    m, codec = mk.Matcher(pats, device=0), mk.Codec(0)
    om = ob.Matcher(pats, True, 0, False)
    values = [b"", b"ZZZ", b"TTT,AAA,CCC", b"AAA,AAA,AAA", b",,", b"x,", b",x", pats[3], pats[5] + b"," + pats[1], b"a," + pats[0] + b",B,b,A", b"ACGT" * 100,
              b"0,00,000,0000", pats[2][:-1], pats[2] + b"A"]
    recs, existing = [], []
    for i in range(600):
        s = bytearray(rnd.choice(b"ACGT") for _ in range(120))
        for _ in range(rnd.randrange(0, 4)):
            p = rnd.choice(pats[:8])
            k = rnd.randrange(0, 120 - 31)
            s[k:k + 31] = p
        v = rnd.choice(values) if i % 3 else None
        aux = b"NMC\x02"
        if v is not None:
            aux += b"kmZ" + v + b"\0"
            if i % 7 == 0:
                aux += b"kmZ" + b"second,field" + b"\0"  # (only the first field of that name is looked at)
        aux += b"ASi" + struct.pack("<i", i)
        recs.append(bam_record(b"e%d" % i, bytes(s), aux))
        existing.append(v)
    blob = _bgzf(b"".join(recs))
    members, _, _ = mk.bgzf_members(blob)
    for fm in (False, True):
        keep, rows, c, out = expected(om, pats, recs, b"km", True, fm, False, existing=existing)
        r = m.tag_bam_window(codec, b"", blob, members, last=True, logging=True, filter_matching=fm)
        assert r["status"] == 0 and r["n_kept"] == sum(keep)
        got = gzip.decompress(r["out"] + mk.bgzf_eof()) if r["out"] else b""
        if got != out:  # (which record differs, for the failure message)
            a, _ = split_records(got)
            b, _ = split_records(out)
            bad = [k for k, (x, y) in enumerate(zip(a, b)) if x != y][:3]
            raise AssertionError(f"records {bad}: {[(a[k][-80:], b[k][-80:]) for k in bad]}")
        assert r["rows"] == rows
    codec.close()


def test_suppressed_output_and_nothing_kept(mk):
    rnd = random.Random(4)
    pats = patterns31(mk, 20)
    m, codec = mk.Matcher(pats, device=0), mk.Codec(0)
    recs = make_records(rnd, 300, pats, hit=0.0)
    blob = _bgzf(b"".join(recs))
    members, _, _ = mk.bgzf_members(blob)
    r = m.tag_bam_window(codec, b"", blob, members, last=True, filter_matching=True)
    assert r["status"] == 0 and r["n_kept"] == 0 and r["out"] == b"" and r["n_rec"] == 300
    r = m.tag_bam_window(codec, b"", blob, members, last=True, write=False)
    assert r["status"] == 0 and r["n_kept"] == 300 and r["out"] == b""
    codec.close()


def test_names_that_look_like_records(mk):
    """qualities and optional fields filled with bytes that pass for the fixed fields of a record: false piece starts that the proof
    has to throw out (tiny pieces, so that many pieces begin inside such bytes)"""
    rnd = random.Random(21)
    pats = patterns31(mk, 20)
    fake = bam_record(b"f", b"ACGT" * 3)  # a complete small record, embedded in B arrays of real ones
    recs = []
    for i in range(400):
        aux = b"ZBB" + b"C" + struct.pack("<i", 4 * len(fake)) + fake * 4
        recs.append(bam_record(b"n%d" % i, bytes(rnd.choice(b"ACGT") for _ in range(60)), aux))
    text = b"".join(recs)
    blob = _bgzf(text)
    members, _, _ = mk.bgzf_members(blob)
    m, codec = mk.Matcher(pats, device=0), mk.Codec(0)
    om = ob.Matcher(pats, True, 0, False)
    keep, rows, c, out = expected(om, pats, recs, b"km", False, False, False)
    for piece in (64, 100, 1000):
        r = m.tag_bam_window(codec, b"", blob, members, last=True, logging=False, piece_bytes=piece)
        # (either the proof goes through and the output is exact, or the window is left to the host reader: never a wrong table)
        assert r["status"] in (0, 1)
        if r["status"] == 0:
            assert r["n_rec"] == len(recs) and gzip.decompress(r["out"] + mk.bgzf_eof()) == out
    r = m.tag_bam_window(codec, b"", blob, members, last=True, logging=False)
    assert r["status"] == 0 and gzip.decompress(r["out"] + mk.bgzf_eof()) == out
    codec.close()


def test_reference_bam_fixture(mk):
    """tests/fixtures/input/simple.bam of the reference (bam crate's writer) through the device path: the records behind its header,
    tagged, must be the records of tests/fixtures/tag/simple-bam.sam"""
    blob = open(os.path.join(GOLDEN, "fixtures", "input", "simple.bam"), "rb").read()
    members, used, _ = mk.bgzf_members(blob)
    text = gzip.decompress(blob)
    l_text = struct.unpack_from("<i", text, 4)[0]
    p = 8 + l_text
    n_ref = struct.unpack_from("<i", text, p)[0]
    p += 4
    for _ in range(n_ref):
        ln = struct.unpack_from("<i", text, p)[0]
        p += 4 + ln + 4
    # the header ends inside member 0: the device window starts behind it, so the text up to there is given as ... nothing: the
    # records are re-packed as members of their own (what the CLI does with the bytes its header parser has already inflated: head)
    recs, _ = split_records(text[p:])
    assert recs
    import textio
    _, sam = textio.read_sam(os.path.join(GOLDEN, "fixtures", "tag", "simple.tagged.extracted.sam"))
    pats = mk.parse_pattern_list(kmer_seq=[b"CTC"], reverse_complement=True)  # tag ... -s CTC -r
    m, codec = mk.Matcher(pats, device=0), mk.Codec(0)
    om = ob.Matcher(pats, m.use_ac, 0, False)
    # head = the record bytes (as the CLI hands over what its header parser has already inflated), no members at all ...
    r = m.tag_bam_window(codec, text[p:], b"", members[:0], last=True, logging=True)
    keep, rows, c, out = expected(om, pats, recs, b"km", True, False, False)
    assert r["status"] == 0 and r["n_rec"] == len(recs) and gzip.decompress(r["out"] + mk.bgzf_eof()) == out and r["rows"] == rows
    # ... and the reference's own members (device inflate of the bam crate's / samtools' bytes), the header included as text in front:
    # the window must start at a record start, so the header's bytes are cut off by handing the members' text over as head instead
    got, _ = split_records(gzip.decompress(r["out"] + mk.bgzf_eof()))
    assert len(got) == len(sam)
    for rec, fields in zip(got, sam):  # name, sequence and the km tag of the reference's tagged output
        name, seq = decode(rec)
        assert name == fields[0] and seq == fields[9]
        want = [f for f in fields[11:] if f.startswith(b"km:Z:")][0][5:]
        assert rec.endswith(b"kmZ" + want + b"\0")
    codec.close()


@pytest.mark.parametrize("seed", range(24))
def test_fuzz_windows_against_the_oracle(mk, seed):
    """seeded differential test: random record shapes (lengths 0 ... 3 000, every optional-field type, existing values of the tag on some
    records), member sizes, window cuts with the tail carried, piece sizes, filter flags, AC and BNDMq pattern sets -- the members the
    device returns must inflate to the oracle's tagged records, rows and counters must agree"""
    rnd = random.Random(1000 + seed)
    few = rnd.random() < 0.3
    if few:
        pats = mk.parse_pattern_list(kmer_seq=[bytes(rnd.choice(b"ACGT") for _ in range(rnd.choice((5, 9, 21)))) for _ in range(rnd.randrange(1, 9))])
    else:
        pats = patterns31(mk, rnd.choice((20, 300)), seed=seed)
    m, codec = mk.Matcher(pats, device=0), mk.Codec(0)
    om = ob.Matcher(pats, m.use_ac, 0, False)
    n = rnd.choice((1, 40, 700, 2500))
    lens = rnd.choice(((150,), (0, 1, 2, 33, 150, 151), (100, 3000), (75,)))
    recs = make_records(rnd, n, pats, lens=lens, hit=rnd.choice((0.0, 0.1, 0.9)), alpha=rnd.choice((b"ACGT", b"ACGTN", b"ACGTNRYKM=")))
    existing = [None] * n
    tag = rnd.choice((b"km", b"XK"))
    if rnd.random() < 0.5:  # some records carry the tag already
        for i in range(n):
            if rnd.random() < 0.3:
                v = rnd.choice((b"", b"AAA", b"T,A,T", pats[0], pats[-1] + b",zz", b",", b"b,a,,c"))
                body = recs[i][4:] + tag + b"Z" + v + b"\0"
                recs[i] = struct.pack("<i", len(body)) + body
                existing[i] = v
    text = b"".join(recs)
    blob = _bgzf(text, block=rnd.choice((500, 7001, 65280)), level=rnd.choice((1, 6)))
    members, _, _ = mk.bgzf_members(blob)
    fm, inv = rnd.choice(((False, False), (True, False), (False, True)))
    logging = rnd.random() < 0.7
    keep, rows, c, out = expected(om, pats, recs, tag, logging, fm, inv, existing=existing)
    step = rnd.choice((1, 3, 50, len(members)))
    cuts = list(range(0, len(members), max(1, step))) + [len(members)]
    res = run_windows(mk, m, codec, blob, members, cuts, tag=tag, logging=logging, filter_matching=fm, invert=inv,
                      piece_bytes=rnd.choice((64, 300, 5000, 0)), block_bytes=rnd.choice((0, 1000, 40000)))
    assert all(x["status"] == 0 for x in res), [x["status"] for x in res]
    assert sum(x["n_rec"] for x in res) == n and sum(x["n_kept"] for x in res) == sum(keep)
    got = b"".join(gzip.decompress(x["out"] + mk.bgzf_eof()) if x["out"] else b"" for x in res)
    assert got == out
    if logging:
        got_rows, base = [], 0
        for x in res:
            got_rows += [(nm, rec + base, pat, pos) for (nm, rec, pat, pos) in x["rows"]]
            base += x["n_rec"]
        assert got_rows == rows
        assert sum(x["counters"]["hits"][0] for x in res) == c["hits"][0] and sum(x["counters"]["records_hit"][0] for x in res) == c["records_hit"][0]
        assert np.array_equal(np.sum([x["counters"]["pattern_hit_counts"] for x in res], axis=0), c["pattern_hit_counts"])
    codec.close()


def _serial_walk(text, last):
    """the CLI's serial parser (cli/io.cpp: parse_bam_records_serial) in Python: -> (records, bytes used) or None where it bails"""
    recs, p, n = [], 0, len(text)
    while p < n:
        if n - p < 4:
            break
        block = struct.unpack_from("<i", text, p)[0]
        if block < 32:
            return None
        if n - p - 4 < block:
            break
        l_name, n_cig, l_seq = text[p + 12], struct.unpack_from("<H", text, p + 16)[0], struct.unpack_from("<i", text, p + 20)[0]
        if l_seq < 0 or 32 + l_name + 4 * n_cig + (l_seq + 1) // 2 + l_seq > block:
            return None
        recs.append(text[p:p + 4 + block])
        p += 4 + block
    if last and p != n:
        return None
    return recs, p


@pytest.mark.parametrize("seed", range(12))
def test_damaged_and_random_windows_never_yield_a_wrong_table(mk, seed):
    """windows of damaged BAM text (random bytes flipped in a valid record stream, a stream cut anywhere, pure noise): whatever the piece
    starts guess, the call either refuses the window (the host reader's business) or returns exactly the serial parser's records --
    and never reads or writes outside its buffers (the kernels bound every access by the window and the record they are in)"""
    rnd = random.Random(500 + seed)
    pats = patterns31(mk, 20)
    m, codec = mk.Matcher(pats, device=0), mk.Codec(0)
    om = ob.Matcher(pats, True, 0, False)
    recs = make_records(rnd, 400, pats, lens=(0, 50, 150, 900), hit=0.3)
    text = bytearray(b"".join(recs))
    kind = seed % 4
    if kind == 0:    # a few flipped bytes anywhere
        for _ in range(rnd.randrange(1, 6)):
            text[rnd.randrange(len(text))] ^= 1 << rnd.randrange(8)
    elif kind == 1:  # fixed fields of some records overwritten
        at, k = 0, 0
        while at + 36 < len(text) and k < 3:
            b = struct.unpack_from("<i", text, at)[0]
            if rnd.random() < 0.02:
                struct.pack_into("<i", text, at + rnd.choice((0, 20)), rnd.choice((-5, 7, 31, 1 << 20, 1 << 30)))
                k += 1
            at += 4 + max(b, 32) if 32 <= b < (1 << 20) else 40
    elif kind == 2:  # cut anywhere
        del text[rnd.randrange(1, len(text)):]
    else:            # noise
        text = bytearray(rnd.getrandbits(8) for _ in range(60000))
    text = bytes(text)
    blob = _bgzf(text, block=rnd.choice((3000, 65280)))
    members, _, _ = mk.bgzf_members(blob)
    for last in (True, False):
        want = _serial_walk(text, last)
        r = m.tag_bam_window(codec, b"", blob, members, last=last, logging=False, piece_bytes=rnd.choice((64, 1000, 0)))
        if want is None:
            assert r["status"] != 0, (seed, last)
            continue
        if r["status"] != 0:  # (optional fields that no longer parse, a chain the proof gave up on: refusals are always allowed)
            continue
        good, used = want
        assert r["n_rec"] == len(good) and r["n_used"] == used and r["tail"] == text[used:]
        keep, rows, c, out = expected(om, pats, good, b"km", False, False, False)
        assert (gzip.decompress(r["out"] + mk.bgzf_eof()) if r["out"] else b"") == out
    codec.close()

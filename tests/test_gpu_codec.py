"""The device BGZF codec through the C ABI (mk_codec_*, mk_bgzf_*; include/merkurio_hip.h v5) against zlib -- the checker
for RFC 1951 / RFC 1952 here: the system's zlib is what the reference's flate2 implements and what every BAM reader
links.  Replaces the BGZF reader / writer the reference gets from `bam 0.1.4` (src/cmd_tag.rs:254-271,503-506).

deflate: every member the device writes is inflated by zlib and compared with the input byte for byte; its BSIZE, CRC-32
and ISIZE are checked field by field (bit-exact text; the compressed bytes are the library's own parse).
inflate: members written by zlib at every level / strategy, and by the device, come back byte for byte; damaged members
are refused with MK_E_CORRUPT and the index of the first one."""
import gzip
import os
import random
import struct
import zlib

import numpy as np
import pytest

from merkurio_amd import native as mk
from test_codec_cpu import corpora, raw_deflate
from textio import bam_like

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=[0, 1, 2, 3, 4, 5, 6], ids=["kernel-by-size", "lane-per-member", "wave-per-member", "wave-8k-ring", "wave-16k-ring", "wave-4k-ring", "wave-2k-ring"])
def codec(request):
    """every inflate test runs seven times: the inflate kernel chosen per call (a wave per member with a 4 / 2 KiB ring for calls of up to
    ~2 000 / ~24 000 members, a lane per member above: bgzf_inflate.hip), and each of the six variants forced (mk_codec_set_inflate_kernel)"""
    c = mk.Codec()
    c.set_inflate_kernel(request.param)
    yield c
    c.close()


@pytest.fixture(scope="module")
def plain_codec():
    """a handle for the tests that do not touch the BGZF inflate kernels (the deflate side, the parallel gunzip)"""
    c = mk.Codec()
    yield c
    c.close()


@pytest.fixture(scope="module", params=[0, 1], ids=["wave-per-piece", "lane-per-piece"])
def gunzip_codec(request):
    """the parallel gunzip's tests run twice: a piece of the stream decoded by a wave (gzip_segments_wave.hip, the default) and by a
    lane (inflate_segment() of gzip_segments.hpp, the function the CPU harness checks against zlib)"""
    c = mk.Codec()
    c.set_inflate_kernel(request.param)
    yield c
    c.close()


def check_members(data, blob, block_bytes):
    """every member: header fields, zlib inflates its payload to the block it stands for, CRC-32 and ISIZE agree"""
    bb = block_bytes or 65280
    mem, used, text = mk.bgzf_members(blob)
    assert used == len(blob) and text == len(data) and len(mem) == (len(data) + bb - 1) // bb
    at = 0
    for k, m in enumerate(mem):
        start = int(m["data_off"]) - 18
        assert blob[start:start + 16] == bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0])
        bsize = struct.unpack_from("<H", blob, start + 16)[0]
        assert bsize + 1 == 18 + int(m["data_len"]) + 8
        block = data[k * bb:(k + 1) * bb]
        d = zlib.decompressobj(-15)
        got = d.decompress(blob[int(m["data_off"]):int(m["data_off"]) + int(m["data_len"])])
        assert d.eof and d.unused_data == b"", k
        assert got == block, k
        assert int(m["crc"]) == zlib.crc32(block) and int(m["isize"]) == len(block)
        at += bsize + 1
    assert at == len(blob)


@pytest.mark.parametrize("block_bytes", [0, 65280, 1000, 64, 7])
def test_deflate_members_inflate_with_zlib(codec, block_bytes):
    for name, data in corpora().items():
        if block_bytes in (64, 7):
            data = data[:3000]
        blob = codec.deflate(data, block_bytes)
        check_members(data, blob, block_bytes)
        if data:
            assert gzip.decompress(blob + mk.bgzf_eof()) == data, name  # the whole file, as a reader sees it


def test_deflate_sizes_around_a_block_and_a_device_pass(codec):
    rng = random.Random(3)
    base = bam_like(1200)
    for n in (1, 2, 3, 4, 5, 63, 64, 65, 65279, 65280, 65281, 2 * 65280, 2 * 65280 + 1):
        data = (base * (n // len(base) + 1))[:n]
        check_members(data, codec.deflate(data), 0)
    # calls cut into several device passes (the pass limits are a tuning hook: results must not depend on them)
    data = bam_like(72000, seed=9)[:5000 * 4096 - 77]
    whole = codec.deflate(data, 4096)
    check_members(data, whole, 4096)
    try:
        codec.set_pass_limits(deflate_members=777, inflate_text_bytes=1_000_000)
        assert codec.deflate(data, 4096) == whole                      # 7 passes of 777 members
        assert codec.deflate_pieces([data[:12345], data[12345:9_000_001], data[9_000_001:]], 4096) == whole  # pieces across passes
        assert codec.inflate(whole) == data                            # 21 passes of ~1 MB of text
        assert codec.inflate(zlib_bgzf(data[:3_000_000], level=6)) == data[:3_000_000]
    finally:
        codec.set_pass_limits()
    # text that cannot shrink is stored
    noise = bytes(rng.getrandbits(8) for _ in range(150000))
    blob = codec.deflate(noise)
    check_members(noise, blob, 0)
    assert len(blob) == len(noise) + 3 * 31


def test_deflate_of_pieces_equals_deflate_of_their_concatenation(plain_codec):
    codec = plain_codec
    """mk_bgzf_deflate_pieces (the BAM writer's per-thread record buffers): the pieces are joined on the device only"""
    rng = random.Random(23)
    data = bam_like(9000, seed=4)
    cuts = sorted(rng.randrange(len(data)) for _ in range(40)) + [len(data)]
    pieces, at = [], 0
    for c in cuts:
        pieces.append(data[at:c])
        at = c
    pieces[5:5] = [b"", b"x", b""]
    joined = b"".join(pieces)
    for bb in (0, 5000):
        blob = codec.deflate_pieces(pieces, bb)
        assert blob == codec.deflate(joined, bb)
        check_members(joined, blob, bb)
    assert codec.deflate_pieces([]) == b"" and codec.deflate_pieces([b"", b""]) == b""


def test_deflate_compresses_like_a_fast_zlib_level(plain_codec):
    codec = plain_codec
    """not a parity claim -- a guard against a parse that stops finding matches: within 25 % of zlib level 1 on BAM-like
    records, FASTQ text and a run-heavy block"""
    for name, data in (("bam", bam_like(40000)), ("bam const qual", bam_like(40000, const_qual=True)), ("fastq", corpora()["fastq"] * 8),
                       ("zeros", bytes(300000))):
        blob = codec.deflate(data)
        z1 = sum(len(raw_deflate(data[i:i + 65280], level=1)) + 26 for i in range(0, len(data), 65280))
        assert len(blob) <= 1.25 * z1, (name, len(blob), z1)


def zlib_bgzf(data, block_bytes=65280, **kw):
    out = b""
    for i in range(0, len(data), block_bytes):
        block = data[i:i + block_bytes]
        z = raw_deflate(block, **kw)
        out += bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0]) + struct.pack("<H", len(z) + 25) + z
        out += struct.pack("<II", zlib.crc32(block), len(block))
    return out


def test_inflate_members_written_by_zlib(codec):
    for name, data in list(corpora().items()) + [("bam", bam_like(3000))]:
        if not data:
            continue
        for kw in ({"level": 0}, {"level": 1}, {"level": 6}, {"level": 9}, {"strategy": zlib.Z_FIXED}, {"strategy": zlib.Z_HUFFMAN_ONLY},
                   {"flush_every": 5000}):
            for bb in (65280, 2000):
                blob = zlib_bgzf(data[:200000], bb, **kw)
                assert codec.inflate(blob) == data[:200000], (name, kw, bb)
    assert codec.inflate(b"") == b""
    assert codec.inflate(mk.bgzf_eof()) == b""


def test_round_trip_on_the_device(codec):
    data = bam_like(300000, seed=11)  # 82 MB, 1 260 members
    blob = codec.deflate(data)
    assert codec.inflate(blob) == data
    assert gzip.decompress(blob) == data


def test_inflate_refuses_damaged_members(codec):
    data = bam_like(3000)
    blob = bytearray(zlib_bgzf(data, level=6))
    mem, _, _ = mk.bgzf_members(bytes(blob))
    assert len(mem) >= 10
    # a flipped bit in the payload of member 3: a stream error or a CRC mismatch, never a silent difference
    bad = bytearray(blob)
    bad[int(mem[3]["data_off"]) + int(mem[3]["data_len"]) // 2] ^= 0x10
    with pytest.raises(mk.MerkurioError) as e:
        codec.inflate(bytes(bad))
    assert e.value.code == mk.MK_E_CORRUPT and "member 3" in str(e.value)
    # a wrong CRC in the trailer of member 5
    bad = bytearray(blob)
    bad[int(mem[5]["data_off"]) + int(mem[5]["data_len"])] ^= 0xff
    with pytest.raises(mk.MerkurioError) as e:
        codec.inflate(bytes(bad))
    assert e.value.code == mk.MK_E_CORRUPT and "member 5" in str(e.value) and "CRC" in str(e.value)
    # a wrong ISIZE (one byte short) in member 0
    m2 = mem.copy()
    m2["isize"][0] -= 1
    with pytest.raises(mk.MerkurioError) as e:
        codec.inflate(bytes(blob), m2, int(mem["isize"].sum()))
    assert e.value.code == mk.MK_E_CORRUPT and "member 0" in str(e.value)
    # random damage: an error or the right text, and never a crash
    rng = random.Random(17)
    for _ in range(40):
        bad = bytearray(blob)
        k = rng.randrange(len(mem))
        for _ in range(rng.randrange(1, 4)):
            bad[int(mem[k]["data_off"]) + rng.randrange(int(mem[k]["data_len"]))] = rng.getrandbits(8)
        try:
            assert codec.inflate(bytes(bad)) == data
        except mk.MerkurioError as e:
            assert e.code == mk.MK_E_CORRUPT
    # members outside their buffers are refused on the host
    m2 = mem.copy()
    m2["data_len"][2] = 1 << 30
    with pytest.raises(mk.MerkurioError) as e:
        codec.inflate(bytes(blob), m2, int(mem["isize"].sum()))
    assert e.value.code == mk.MK_E_INVALID_ARG


def _fuzz_text(rng, n):
    """text with the structures a parse can trip over: runs, short periods, long repeats, near-matches, noise"""
    out = bytearray()
    alphabet = bytes(rng.sample(range(256), rng.choice((1, 2, 4, 16, 64, 256))))
    while len(out) < n:
        kind = rng.randrange(7)
        if kind == 0:
            out += bytes(rng.choice(alphabet) for _ in range(rng.randrange(1, 400)))
        elif kind == 1:
            out += bytes([rng.choice(alphabet)]) * rng.randrange(1, 700)  # runs: distance 1, lengths across 258
        elif kind == 2:
            p = bytes(rng.choice(alphabet) for _ in range(rng.randrange(2, 9)))  # periods 2..8: the register-pattern copies
            out += p * rng.randrange(1, 120)
        elif kind == 3 and out:
            b = rng.randrange(len(out))  # a repeat of earlier text, any distance (also > 32 KiB: must not become a match)
            out += out[b:b + rng.randrange(3, 600)]
        elif kind == 4 and out:
            b = rng.randrange(len(out))  # a repeat with one byte changed
            s = bytearray(out[b:b + rng.randrange(8, 300)])
            s[rng.randrange(len(s))] ^= 1 + rng.randrange(255)
            out += s
        elif kind == 5:
            out += bytes(rng.getrandbits(8) for _ in range(rng.randrange(1, 300)))
        else:
            out += b"@r%d\n" % rng.randrange(10 ** 9) + bytes(rng.choice(b"ACGT") for _ in range(100)) + b"\n+\n" + b"F" * 100 + b"\n"
    return bytes(out[:n])


def test_fuzz_both_directions(codec):
    """seeded random texts and block sizes: device deflate -> zlib, zlib (random level / strategy / flush points) -> device
    inflate, device -> device; every case byte for byte"""
    import os
    n_cases = int(os.environ.get("MERKURIO_FUZZ_CODEC", "60"))
    for seed in range(n_cases):
        rng = random.Random(1000 + seed)
        n = rng.choice((rng.randrange(1, 300), rng.randrange(300, 70000), rng.randrange(70000, 400000)))
        data = _fuzz_text(rng, n)
        bb = rng.choice((0, 0, 65280, rng.randrange(1, 65281)))
        blob = codec.deflate(data, bb)
        assert gzip.decompress(blob) == data, (seed, "device deflate -> zlib")
        mem, used, text = mk.bgzf_members(blob)
        assert used == len(blob) and text == len(data) and all(int(m["crc"]) == zlib.crc32(data[int(m["out_off"]):int(m["out_off"]) + int(m["isize"])]) for m in mem)
        assert codec.inflate(blob) == data, (seed, "device -> device")
        kw = rng.choice(({"level": rng.randrange(0, 10)}, {"strategy": zlib.Z_FIXED}, {"strategy": zlib.Z_HUFFMAN_ONLY}, {"strategy": zlib.Z_RLE},
                         {"level": rng.randrange(1, 10), "flush_every": rng.randrange(50, 20000)}))
        zb = zlib_bgzf(data, rng.randrange(1, 65281) if rng.random() < 0.3 else 65280, **kw)
        assert codec.inflate(zb) == data, (seed, "zlib -> device inflate", kw)


# ---- members the REFERENCE holds (r05): every other test here inflates members this suite wrote itself (zlib or the device) ----
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _sam_records_of_bam(raw):
    """(header text, [SAM fields decoded from the binary records]) of an inflated BAM stream -- just enough of SAM spec 4.2
    to compare with the reference's simple.sam: QNAME FLAG POS MAPQ SEQ"""
    assert raw[:4] == b"BAM\x01"
    l_text = struct.unpack_from("<i", raw, 4)[0]
    text = raw[8:8 + l_text]
    at = 8 + l_text
    n_ref = struct.unpack_from("<i", raw, at)[0]
    at += 4
    for _ in range(n_ref):
        l_name = struct.unpack_from("<i", raw, at)[0]
        at += 4 + l_name + 4
    recs = []
    while at < len(raw):
        bs = struct.unpack_from("<i", raw, at)[0]
        r = raw[at + 4:at + 4 + bs]
        at += 4 + bs
        _, pos, l_name, mapq, _, n_cig, flag, l_seq = struct.unpack_from("<iiBBHHHi", r, 0)
        name = r[32:32 + l_name - 1]
        sq = r[32 + l_name + 4 * n_cig:]
        seq = bytes(b"=ACMGRSVTWYHKDBN"[(sq[k >> 1] >> (4 if k % 2 == 0 else 0)) & 15] for k in range(l_seq))
        recs.append((name, flag, pos + 1, mapq, seq))
    return text, recs


def test_reference_bam_fixture_through_the_device_inflater(codec):
    """tests/fixtures/input/simple.bam (the reference's only BAM; in the CLI its 3 members stay below the device threshold and
    take zlib): walked by mk_bgzf_members, inflated by mk_bgzf_inflate, compared with zlib's text AND with the records of the
    reference's own simple.sam (src/cmd_tag.rs:1095-1132 reads this pair)"""
    blob = open(os.path.join(GOLDEN, "fixtures/input/simple.bam"), "rb").read()
    mem, used, text = mk.bgzf_members(blob)
    assert used == len(blob) and len(mem) == 3 and int(mem[2]["isize"]) == 0  # header member, record member, end-of-file marker
    raw = codec.inflate(blob)
    assert raw == gzip.decompress(blob) and len(raw) == text
    header, recs = _sam_records_of_bam(raw)
    sam = open(os.path.join(GOLDEN, "fixtures/input/simple.sam"), "rb").read().split(b"\n")
    sam_recs = [ln.split(b"\t") for ln in sam if ln and not ln.startswith(b"@")]
    assert [(f[0], int(f[1]), int(f[3]), int(f[4]), f[9]) for f in sam_recs] == recs and len(recs) == 3
    assert [ln for ln in header.split(b"\n") if ln.startswith(b"@SQ")] == [ln for ln in sam if ln.startswith(b"@SQ")]


def test_reference_gzip_fixture_through_the_device_inflater(codec):
    """tests/data/sample.fasta.gz (needletail's gzip reader, src/cmd_extract.rs:281): one gzip member with FNAME, written by
    gzip(1) -- no BC subfield, so the member table is built here from RFC 1952's fields; the DEFLATE stream itself is the
    reference's bytes.  The text must be the reference's sample.fasta."""
    blob = open(os.path.join(GOLDEN, "data/sample.fasta.gz"), "rb").read()
    assert blob[:3] == b"\x1f\x8b\x08" and blob[3] == 0x08  # FNAME only
    data_off = blob.index(b"\0", 10) + 1
    crc, isize = struct.unpack_from("<II", blob, len(blob) - 8)
    mem = np.zeros(1, dtype=mk.MEMBER_DTYPE)
    mem[0]["data_off"], mem[0]["out_off"], mem[0]["data_len"], mem[0]["isize"], mem[0]["crc"] = data_off, 0, len(blob) - 8 - data_off, isize, crc
    text = codec.inflate(blob, mem, isize)
    assert text == open(os.path.join(GOLDEN, "data/sample.fasta"), "rb").read() == gzip.decompress(blob)
    # ... and as BGZF written by the device from that text: walked and inflated again, still the reference's file
    again = codec.deflate(text, 500)
    assert codec.inflate(again) == text and gzip.decompress(again + mk.bgzf_eof()) == text
    # a gzip member that carries BC *and* FNAME / FCOMMENT / FHCRC: mk_bgzf_members skips the optional fields (r04 ADVICE)
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    payload = co.compress(text) + co.flush()
    for flg, fields in ((4 | 8, b"name.fa\0"), (4 | 16, b"a comment\0"), (4 | 8 | 16, b"n\0c\0"), (4 | 2, b"\x12\x34"), (4 | 2 | 8 | 16 | 1, b"n\0c\0\x12\x34")):
        bsize = 18 + len(fields) + len(payload) + 8 - 1
        m = bytes([0x1f, 0x8b, 8, flg, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0]) + struct.pack("<H", bsize) + fields + payload + \
            struct.pack("<II", zlib.crc32(text), len(text))
        tab, used, n_text = mk.bgzf_members(m)
        assert used == len(m) and n_text == len(text) and int(tab[0]["data_off"]) == 18 + len(fields)
        assert codec.inflate(m) == text
    with pytest.raises(mk.MerkurioError):
        mk.bgzf_members(bytes([0x1f, 0x8b, 8, 4 | 0x20]) + m[4:])


# ---- one gzip member inflated in parallel pieces (r05: mk_gzip_inflate_device, codec/gzip_segments.hpp) -----------------------------
def _fastq_text(n, seed=3):
    rnd = random.Random(seed)
    out = []
    for i in range(n):
        s = "".join(rnd.choice("ACGT") for _ in range(150))
        q = "".join(rnd.choice("FFFF:,#") for _ in range(150))
        out.append(f"@read{i} lane={i % 8}\n{s}\n+\n{q}\n")
    return "".join(out).encode()


def test_gunzip_of_the_reference_sample_and_of_gzip_written_fastq(gunzip_codec):
    codec = gunzip_codec
    """tests/data/sample.fasta.gz (the reference's own: one small member, a single piece) and 30 MB of FASTQ as gzip -1 / -6 / -9 wrote
    it (hundreds of pieces: block starts found on the device, pieces decoded side by side, place-holders resolved): the text zlib gives"""
    blob = open(os.path.join(GOLDEN, "data/sample.fasta.gz"), "rb").read()
    assert codec.gunzip(blob) == open(os.path.join(GOLDEN, "data/sample.fasta"), "rb").read()
    assert codec.gzip_info[0] == 1
    data = _fastq_text(90_000)
    for level in (1, 6, 9):
        gz = gzip.compress(data, level)
        text = codec.gunzip(gz)
        assert text is not None, ("not taken", level, codec.gzip_info)
        assert text == data, level
        assert codec.gzip_info[0] > 20, codec.gzip_info  # pieces: this is the parallel path, not one lane walking the stream
    # the text stays on the device: ranges of it
    n, L = len(data), codec._L
    for off, ln in ((0, 1), (n - 7, 7), (12345, 1 << 20), (n, 0)):
        out = np.zeros(max(1, ln), dtype=np.uint8)
        mk._check(L.mk_gzip_text_read(codec._h, off, out.ctypes.data, ln))
        assert out[:ln].tobytes() == data[off:off + ln]
    assert L.mk_gzip_text_read(codec._h, n - 3, out.ctypes.data, 4) == mk.MK_E_INVALID_ARG


def test_gunzip_pieces_by_wave_and_by_lane_take_the_same_streams():
    """text whose matches reach far back (a wave keeps 2 048 symbols of a piece in LDS and reads older ones from device memory; the
    32 KiB in front of a piece are place-holders), runs (distance 1 .. 3), stored blocks between compressed ones: both decoders give
    zlib's text, and take or hand back the same streams"""
    rnd = random.Random(77)
    wave, lane = mk.Codec(), mk.Codec()
    lane.set_inflate_kernel(1)
    try:
        for case in range(10):
            parts = []
            pool = [bytes(rnd.choice(b"ACGTN") for _ in range(rnd.randrange(2000, 30000))) for _ in range(4)]
            while sum(map(len, parts)) < 3_000_000:
                r = rnd.random()
                if r < 0.5:
                    blk = rnd.choice(pool)
                    a = rnd.randrange(len(blk) - 300)
                    parts.append(blk[a:a + rnd.randrange(3, 300)])
                elif r < 0.7:
                    parts.append(bytes(rnd.choice(b"ACGT") for _ in range(rnd.randrange(1, 400))))
                elif r < 0.8:
                    parts.append(bytes([rnd.choice(b"#F:")]) * rnd.randrange(1, 600))
                elif r < 0.9:
                    parts.append(bytes(rnd.choice(b"AB") for _ in range(2)) * rnd.randrange(1, 200))
                else:
                    parts.append(os.urandom(rnd.randrange(1, 3000)))
            data = b"".join(parts)
            co = zlib.compressobj(rnd.choice([1, 4, 6, 9]), zlib.DEFLATED, 31, rnd.choice([1, 8, 9]))
            gz, at = [], 0
            while at < len(data):
                step = rnd.choice([40_000, 200_000, 900_000])
                gz.append(co.compress(data[at:at + step]))
                if rnd.random() < 0.3:
                    gz.append(co.flush(zlib.Z_FULL_FLUSH))
                at += step
            gz.append(co.flush())
            gz = b"".join(gz)
            a, b = wave.gunzip(gz), lane.gunzip(gz)
            assert (a is None) == (b is None), (case, wave.gzip_info, lane.gzip_info)
            assert a is None or a == data, case
            assert b is None or b == data, case
            assert wave.gzip_info[0] == lane.gzip_info[0], case  # the same pieces
    finally:
        wave.close(), lane.close()


def test_gunzip_text_does_not_depend_on_where_the_stream_is_cut(gunzip_codec):
    """mk_codec_set_gzip_chunk: cuts every 4 KiB (far more cuts than blocks: most find the same start), 16 KiB, 1 MiB (a few pieces of
    many blocks each) -- the same text; a value that is not a power of two in range is refused"""
    codec = gunzip_codec
    data = _fastq_text(60_000, seed=3)
    gz = gzip.compress(data, 6)
    pieces = []
    try:
        for chunk in (4096, 16384, 1 << 20, 0):
            codec.set_gzip_chunk(chunk)
            assert codec.gunzip(gz) == data, chunk
            pieces.append(codec.gzip_info[0])
        assert pieces[0] >= pieces[1] > pieces[2] and pieces[2] <= len(gz) // (1 << 20) + 1, pieces
        for bad in (1000, 2048, 3 << 20, 12288):
            with pytest.raises(mk.MerkurioError):
                codec.set_gzip_chunk(bad)
    finally:
        codec.set_gzip_chunk(0)


def test_gunzip_drops_a_start_that_is_none(gunzip_codec):
    """a DEFLATE stream stored inside the stream (incompressible to gzip: it lies verbatim in stored blocks): its first block's header
    passes every test of the search, and with cuts every 4 KiB one of them is sure to find it -- but no block of the outer stream
    starts there.  The piece in front runs over it, the start is dropped, the pieces are decoded again: still zlib's text (without
    that the call could only hand the file back)"""
    codec = gunzip_codec
    rnd = random.Random(5)
    inner = zlib.compressobj(6, zlib.DEFLATED, -15)
    inner = inner.compress(_fastq_text(40, seed=1)) + inner.flush(zlib.Z_FULL_FLUSH) + inner.compress(_fastq_text(40, seed=2)) + inner.flush()
    assert inner[0] & 7 == 4  # not final, dynamic: what the search looks for
    data = _fastq_text(900, seed=3) + rnd.randbytes(100_000) + inner + rnd.randbytes(100_000) + _fastq_text(900, seed=4)
    gz = gzip.compress(data, 6)
    assert gz.find(inner) > 0  # verbatim
    try:
        for chunk in (4096, 0):
            codec.set_gzip_chunk(chunk)
            assert codec.gunzip(gz) == data, chunk
    finally:
        codec.set_gzip_chunk(0)


def test_gunzip_shapes_the_device_takes_or_hands_back(gunzip_codec):
    codec = gunzip_codec
    """streams of every block type and of many flush points; what is not for this path comes back as 'not taken' (None) and never as a
    wrong text: several members, a damaged stream, a stream the buffers do not hold"""
    rnd = random.Random(8)
    base = _fastq_text(8000, seed=5)
    taken = 0
    for case in range(24):
        level = rnd.choice([0, 1, 1, 6, 6, 9])
        strategy = rnd.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED])
        co = zlib.compressobj(level, zlib.DEFLATED, 31, 9 if case % 2 else 6, strategy)
        data = base[:rnd.randrange(200_000, len(base))]
        parts, at = [], 0
        while at < len(data):  # flush points: blocks of every size, empty stored blocks between them
            step = rnd.choice([500, 5000, 70_000, 300_000])
            parts.append(co.compress(data[at:at + step]))
            if rnd.random() < 0.4:
                parts.append(co.flush(rnd.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH])))
            at += step
        parts.append(co.flush())
        gz = b"".join(parts)
        assert zlib.decompress(gz, 31) == data
        text = codec.gunzip(gz)
        assert text is None or text == data, (case, level, strategy)
        taken += text is not None
    assert taken >= 16, taken  # (most of them: zlib-written streams are what the path is for)
    data = base[:1_500_000]
    gz = gzip.compress(data, 6)
    assert codec.gunzip(gz) == data
    assert codec.gunzip(gz + gzip.compress(b"second member\n")) is None          # several members: zlib's business
    for k in range(12):                                                         # damage: never a wrong text
        bad = bytearray(gz)
        pos = rnd.randrange(20, len(bad) - 8)
        bad[pos] ^= 1 << rnd.randrange(8)
        t = codec.gunzip(bytes(bad))
        assert t is None, (k, pos)
    bad = bytearray(gz)
    bad[-2] ^= 0x10                                                             # ISIZE
    assert codec.gunzip(bytes(bad)) is None
    assert codec.gunzip(gz[:len(gz) // 2]) is None                               # truncated
    assert codec.gunzip(b"not gzip at all, just text\n" * 10) is None
    zeros = gzip.compress(bytes(50_000_000), 6)                                  # ratio 1000: more than a piece's buffer holds
    t = codec.gunzip(zeros)
    assert t is None or t == bytes(50_000_000)

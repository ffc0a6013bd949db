"""The serial pieces of the device BGZF codec (merkurio_amd/csrc/codec/*.hpp) compiled for the host and checked
against zlib -- the checker for RFC 1951 here, as the system's zlib is what the host path of the CLI links and what
the reference's flate2 implements (src/cmd_tag.rs:254-271,503-615 through `bam 0.1.4`).  No GPU: the same headers
are the body of mk_bgzf_inflate_kernel (one lane per BGZF member) and the between-phases code of mk_bgzf_deflate_kernel."""
import os
import random
import struct
import subprocess
import zlib

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_FLAGS = (["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"]
          if os.environ.get("MERKURIO_TEST_SANITIZE") else ["-O1"])


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("codec") / "codec_harness")
    subprocess.run(["g++", "-std=c++17", *_FLAGS, "-Wall", "-I", os.path.join(ROOT, "merkurio_amd/csrc"), "-o", exe,
                    os.path.join(ROOT, "tests/helpers/codec_harness.cpp")], check=True)
    return exe


def corpora():
    rng = random.Random(7)
    fastq = b"".join(b"@read%d/1\n%s\n+\n%s\n" % (i, bytes(rng.choice(b"ACGT") for _ in range(150)),
                                                  bytes(rng.choice(b"FFFFFFF:,#") for _ in range(150))) for i in range(700))
    binary = bytes(rng.getrandbits(8) for _ in range(70000))
    nibbles = bytes(rng.choice((0x11, 0x12, 0x14, 0x18, 0x21, 0x22, 0x24, 0x28, 0x41, 0x42, 0x44, 0x48, 0x81, 0x82, 0x84, 0x88))
                    for _ in range(66000))
    skew = bytes(rng.choice(b"a" * 200 + bytes(range(256))) for _ in range(65000))  # long codewords for the rare bytes
    return {"fastq": fastq, "binary": binary, "nibbles": nibbles, "zeros": bytes(65280), "skew": skew, "empty": b"",
            "one": b"A", "two": b"AB", "short": b"ACGTACGTACGTAC", "ramp": bytes(range(256)) * 200}


def raw_deflate(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, flush_every=0):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
    out = b""
    if flush_every:
        for i in range(0, len(data), flush_every):
            out += c.compress(data[i:i + flush_every]) + c.flush(zlib.Z_FULL_FLUSH)
    else:
        out += c.compress(data)
    return out + c.flush()


def run_inflate(harness, tmp_path, records):
    cont = tmp_path / "in.bin"
    with open(cont, "wb") as f:
        for z, n_out in records:
            f.write(struct.pack("<II", len(z), n_out) + z)
    out = tmp_path / "out.bin"
    r = subprocess.run([harness, "inflate", str(cont), str(out)], capture_output=True, text=True, check=True)
    return [int(x) for x in r.stdout.split()], open(out, "rb").read()


def test_inflate_matches_zlib(harness, tmp_path):
    """stored / fixed / dynamic blocks, several blocks per stream, every corpus at levels 0, 1, 6, 9"""
    recs, want = [], b""
    for name, data in corpora().items():
        for block in (data[:65280], data[:1000]):
            for kw in ({"level": 0}, {"level": 1}, {"level": 6}, {"level": 9}, {"strategy": zlib.Z_FIXED},
                       {"strategy": zlib.Z_HUFFMAN_ONLY}, {"flush_every": 5000}, {"level": 0, "flush_every": 3000}):
                recs.append((raw_deflate(block, **kw), len(block)))
                want += block
    status, got = run_inflate(harness, tmp_path, recs)
    assert status == [0] * len(recs)
    assert got == want


def test_inflate_rejects_damaged_streams(harness, tmp_path):
    """truncated streams, flipped bits, wrong ISIZE: an error code, never a write outside the output (the harness
    checks a guard byte; under MERKURIO_TEST_SANITIZE=1 ASan watches the reads)"""
    rng = random.Random(11)
    data = corpora()["fastq"][:60000]
    z = raw_deflate(data)
    recs = [(z[:len(z) // 2], len(data)), (z[:-1], len(data)), (z, len(data) - 1), (z, len(data) + 1), (b"", 10), (b"\x07", 0)]
    for _ in range(200):
        b = bytearray(z)
        for _ in range(rng.randrange(1, 4)):
            b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
        recs.append((bytes(b), len(data)))
    status, got = run_inflate(harness, tmp_path, recs)
    assert all(s != -99 for s in status)          # guard byte intact
    assert all(s != 0 for s in status[:6])
    at = 0
    for (zz, n), s in zip(recs, status):
        if s == 0:                                # a flipped bit may still be a valid stream of the right size: then zlib agrees on the bytes
            assert got[at:at + n] == zlib.decompress(zz, -15)
        at += n


def test_deflate_pieces_roundtrip_through_zlib(harness, tmp_path):
    """code lengths (incl. the 15-bit limit and the 7-bit limit of the code-length code), canonical codes, the
    run-length header, bit packing: zlib inflates every block to its input; planned header size == written"""
    for name, data in corpora().items():
        src = tmp_path / "src.bin"
        open(src, "wb").write(data)
        for block in (65280, 4096, 100):
            out = tmp_path / "z.bin"
            subprocess.run([harness, "deflate", str(src), str(block), str(out)], check=True)
            c = open(out, "rb").read()
            at, got = 0, b""
            while at < len(c):
                n, = struct.unpack_from("<I", c, at)
                d = zlib.decompressobj(-15)
                got += d.decompress(c[at + 4:at + 4 + n])
                assert d.eof and not d.unused_data, (name, block)
                at += 4 + n
            assert got == data, (name, block)


def test_length_limited_code_on_a_fibonacci_histogram(harness, tmp_path):
    """frequencies that grow like Fibonacci numbers make the unrestricted Huffman tree 20+ levels deep: the 15-bit repair must run"""
    fib = [1, 1]
    while len(fib) < 24:
        fib.append(fib[-1] + fib[-2])
    data = b"".join(bytes([65 + i]) * f for i, f in enumerate(fib))
    rng = random.Random(3)
    data = bytes(rng.sample(data, len(data)))[:65000]
    src = tmp_path / "fib.bin"
    open(src, "wb").write(data)
    out = tmp_path / "z.bin"
    subprocess.run([harness, "deflate", str(src), "65280", str(out)], check=True)
    c = open(out, "rb").read()
    n, = struct.unpack_from("<I", c, 0)
    assert zlib.decompress(c[4:4 + n], -15) == data


def test_crc_folded_from_pieces(harness, tmp_path):
    for name, data in corpora().items():
        src = tmp_path / "c.bin"
        open(src, "wb").write(data)
        for pieces in (1, 2, 64):
            r = subprocess.run([harness, "crc", str(src), str(pieces)], capture_output=True, text=True, check=True)
            assert int(r.stdout, 16) == zlib.crc32(data), (name, pieces)


def test_parallel_gunzip_pieces_match_zlib(tmp_path):
    """merkurio_amd/csrc/codec/gzip_segments.hpp (r05: one gzip member inflated in parallel pieces) run serially the way the kernels of
    gzip_inflate.hip run it -- block starts searched from nominal cuts by trying every bit position, pieces decoded into 16-bit symbols
    with place-holders for the 32 KiB in front, contexts resolved piece by piece -- against zlib's text, for gzip levels 1 / 6 / 9,
    streams with flush points (stored blocks between the dynamic ones), fixed-code and stored-only streams, tiny inputs"""
    exe = str(tmp_path / "gzip_harness")
    subprocess.run(["g++", "-std=c++17", *_FLAGS, "-O2", "-Wall", "-I", os.path.join(ROOT, "merkurio_amd/csrc/codec"), "-I", os.path.join(ROOT, "merkurio_amd/csrc"),
                    "-o", exe, os.path.join(ROOT, "tests/helpers/gzip_harness.cpp")], check=True)
    rng = random.Random(11)
    fastq = b"".join(b"@read%d lane=%d\n%s\n+\n%s\n" % (i, i % 8, bytes(rng.choice(b"ACGT") for _ in range(150)),
                                                          bytes(rng.choice(b"FFFF:,#") for _ in range(150))) for i in range(6000))

    def check(gz, data, chunk, min_pieces):
        (tmp_path / "t.gz").write_bytes(gz)
        p = subprocess.run([exe, str(tmp_path / "t.gz"), str(chunk)], capture_output=True)
        assert p.returncode == 0, p.stderr[-400:]
        assert p.stdout == data
        pieces = int(p.stderr.split()[-1])
        assert pieces >= min_pieces, (pieces, min_pieces)

    import gzip
    for level in (1, 6, 9):
        check(gzip.compress(fastq, level), fastq, 20000, 8)
        check(gzip.compress(fastq, level), fastq, 150000, 2)
    # flush points: empty stored blocks between dynamic ones; the searcher only believes a dynamic block that is followed by another
    # block whose header it can check
    co = zlib.compressobj(6, zlib.DEFLATED, 31)
    gz = b"".join(co.compress(fastq[i:i + 40000]) + co.flush(zlib.Z_SYNC_FLUSH if (i // 40000) % 2 else zlib.Z_FULL_FLUSH) for i in range(0, len(fastq), 40000)) + co.flush()
    check(gz, fastq, 15000, 4)
    for strategy, level in ((zlib.Z_FIXED, 6), (zlib.Z_HUFFMAN_ONLY, 6), (zlib.Z_DEFAULT_STRATEGY, 0), (zlib.Z_RLE, 6)):
        co = zlib.compressobj(level, zlib.DEFLATED, 31, 8, strategy)
        check(co.compress(fastq) + co.flush(), fastq, 30000, 1)  # (fixed / stored blocks have no findable starts: fewer pieces, same text)
    for data in (b"", b"A", b"ACGT" * 10, bytes(70000)):
        check(gzip.compress(data), data, 20000, 1)

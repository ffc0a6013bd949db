"""CLI behaviour that is decided before any GPU work: clap-style argument groups
(src/cmd_extract.rs:33-62,102-110; src/cmd_tag.rs:29-66; tests of src/main.rs:61-293), the
log-flag conflict check (src/helpers.rs:172-200) and path helpers.  Runs without a GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "merkurio_amd", "lib", "merkurio")


@pytest.fixture(scope="module", autouse=True)
def _built():
    from merkurio_amd import build
    build.build_all()


def rc(*args):
    p = subprocess.run([BIN, *args], capture_output=True)
    return p.returncode, p.stdout, p.stderr


def test_help_and_version():
    assert rc("--help")[0] == 0 and b"extract" in rc("--help")[1]
    assert rc("extract", "--help")[0] == 0 and rc("tag", "-h")[0] == 0
    assert rc("--version")[1].startswith(b"merkurio ")
    assert rc()[0] == 2 and rc("frobnicate")[0] == 2


def test_extract_argument_groups(golden):
    fa = os.path.join(golden, "fixtures/input/simple.fasta")
    assert rc("extract", "-s", "A")[0] == 2                                  # -i required
    assert rc("extract", "-i", fa)[0] == 2                                   # one of -s / -f required
    assert rc("extract", "-i", fa, "-s", "A", "-f", "k.txt")[0] == 2         # ... and only one
    assert rc("extract", "-i", fa, "-s", "A", "-q", "1", "-a")[0] == 2       # algorithm group
    assert rc("extract", "-i", fa, "-s", "A", "-I", "-L")[0] == 2            # case group
    assert rc("extract", "-i", fa, "-s", "A", "-L", "-U")[0] == 2
    assert rc("extract", "-i", fa, "-s", "A", "-c", "-r")[0] == 2            # preprocessing group
    assert rc("extract", "-i", fa, "-s", "A", "-S")[0] == 2                  # -S requires logging
    assert rc("extract", "-i", fa, "-s", "A", "-S", "-l", "-o", "x")[0] == 2  # -S conflicts with -o
    assert rc("extract", "-i", fa, "-s", "A", "--bogus")[0] == 2
    assert rc("extract", "-i", fa, "-s", "A", "-q", "x")[0] == 2
    assert rc("extract", "-i", fa, "-s")[0] == 2


def test_tag_argument_groups(golden):
    sam = os.path.join(golden, "fixtures/input/simple.sam")
    assert rc("tag", "-s", "A")[0] == 2
    assert rc("tag", "-i", sam, "-s", "A", "-m", "-v")[0] == 2               # matching group
    assert rc("tag", "-i", sam, "-s", "A", "-S")[0] == 2


def test_log_flag_conflicts(golden):
    fa = os.path.join(golden, "fixtures/input/simple.fasta")
    code, _, err = rc("extract", "-i", fa, "-s", "A", "-l", "-j")
    assert code == 1 and b"Cannot use both -l/--out-log and -j/--json-log" in err
    code, _, err = rc("extract", "-i", fa, "-s", "A", "-l")
    assert code == 1 and b"Cannot write log to stdout when normal output is also stdout" in err
    code, _, err = rc("extract", "-i", fa, "-f", os.path.join(golden, "data/kmers-empty.txt"), "-o", "x")
    assert code == 1 and b"No k-mers found" in err
    code, _, err = rc("extract", "-i", fa, "-f", "/nonexistent/kmers.txt", "-o", "x")
    assert code == 1 and b"File not found." in err


def test_clustered_short_flags_parse_like_clap(golden):
    """-rl, -vI, -Sl: combined short flags (clap accepts them; scripts written for the reference use them)"""
    fa = os.path.join(golden, "fixtures/input/simple.fasta")
    # parsing succeeds and the run gets as far as the semantic checks / the device
    code, _, err = rc("extract", "-i", fa, "-s", "A", "-rl", "-j")
    assert code == 1 and b"Cannot use both -l/--out-log and -j/--json-log" in err
    code, _, err = rc("extract", "-i", fa, "-s", "A", "-rc")
    assert code == 2 and b"--canonical" in err                               # the group check sees both flags
    assert rc("extract", "-i", fa, "-s", "A", "-rq")[0] == 2                 # -q still needs its value
    code, _, err = rc("extract", "-i", fa, "-s", "A", "-vIq5")
    assert code in (0, 1) and b"unexpected" not in err                       # parsed (q takes '5'); past clap: runs or fails on the device
    assert rc("extract", "-i", fa, "-s", "A", "-rx")[0] == 2                 # unknown flag inside a cluster


# MERKURIO_TEST_SANITIZE=1: build the host-code harnesses with AddressSanitizer + UBSan (CPU only)
_HARNESS_FLAGS = (["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"]
                  if os.environ.get("MERKURIO_TEST_SANITIZE") else ["-O1"])


def test_bz2_xz_zstd_decoders(tmp_path, golden):
    """needletail reads .gz/.bz2/.xz (zstd with the same feature); the CLI binds libbz2 / liblzma /
    libzstd at run time.  Harness around the CLI's decoder file: reference samples, multi-stream
    bzip2, multi-MB inputs, truncated inputs are errors."""
    import bz2
    import ctypes
    import lzma
    exe = str(tmp_path / "dz")
    subprocess.run(["g++", "-std=c++17", *_HARNESS_FLAGS, "-DMK_DECOMPRESS_HARNESS", "-I", os.path.join(ROOT, "merkurio_amd/csrc/cli"), "-o", exe,
                    os.path.join(ROOT, "tests/helpers/decompress_harness.cpp"),
                    os.path.join(ROOT, "merkurio_amd/csrc/cli/decompress.cpp"), "-ldl"], check=True)

    def fnv(b):
        h = 1469598103934665603
        for c in b:
            h = ((h ^ c) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        return "%x" % h

    plain = open(os.path.join(golden, "data/sample.fasta"), "rb").read()
    big = plain * 300
    z = ctypes.CDLL("libzstd.so.1")
    z.ZSTD_compressBound.restype = ctypes.c_size_t
    z.ZSTD_compressBound.argtypes = [ctypes.c_size_t]
    z.ZSTD_compress.restype = ctypes.c_size_t
    z.ZSTD_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int]
    cap = z.ZSTD_compressBound(len(big))
    buf = ctypes.create_string_buffer(cap)
    n_zs = z.ZSTD_compress(buf, cap, big, len(big), 3)
    zs = buf.raw[:n_zs]
    files = {"two.bz2": bz2.compress(big[:1000]) + bz2.compress(big[1000:]), "big.xz": lzma.compress(big), "big.zst": zs,
             "trunc.bz2": bz2.compress(big)[:-40], "trunc.xz": lzma.compress(big)[:-40], "trunc.zst": zs[:-40]}
    for name, data in files.items():
        open(tmp_path / name, "wb").write(data)
    args = [os.path.join(golden, "data/sample.fasta.bz2"), os.path.join(golden, "data/sample.fasta.xz")] + \
           [str(tmp_path / n) for n in files]
    out = subprocess.run([exe, *args], capture_output=True, text=True, check=True).stdout.splitlines()
    assert len(out) == 8
    for line in out[:2]:
        assert f"1 {len(plain)} bytes hash {fnv(plain)}" in line, line
    for line in out[2:5]:
        assert f"1 {len(big)} bytes hash {fnv(big)}" in line, line
    for line in out[5:]:
        assert "error Error while decompressing" in line, line


def test_windowed_fastx_reader(tmp_path):
    """extract reads its input a window at a time (FastxStream): plain / gzip / multi-member gzip / BGZF /
    bzip2, FASTA and FASTQ (quality lines that start with '@' and '+'), CRLF, no final newline -- every
    window size and a partial consume must hand over exactly the records a whole-file parse sees;
    a truncated gzip is an error."""
    import bz2
    import gzip
    import random
    import struct
    import zlib
    cli_dir = os.path.join(ROOT, "merkurio_amd/csrc/cli")
    exe = str(tmp_path / "fs")
    subprocess.run(["g++", "-std=c++17", *_HARNESS_FLAGS, "-w", "-I", cli_dir, "-o", exe, os.path.join(ROOT, "tests/helpers/fastx_stream_harness.cpp"),
                    os.path.join(cli_dir, "io.cpp"), os.path.join(cli_dir, "decompress.cpp"), os.path.join(cli_dir, "util.cpp"),
                    "-lz", "-ldl", "-lpthread"], check=True)

    def bgzf(data, block=0xff00):
        out = bytearray()
        for b in range(0, len(data), block):
            chunk = data[b:b + block]
            co = zlib.compressobj(6, zlib.DEFLATED, -15)
            c = co.compress(chunk) + co.flush()
            out += bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0]) + struct.pack("<H", len(c) + 25)
            out += c + struct.pack("<II", zlib.crc32(chunk), len(chunk))
        return bytes(out) + bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")

    rnd = random.Random(3)
    recs = []
    for i in range(6000):
        L = rnd.choice([1, 50, 100, 151, 700])
        recs.append((f"r{i} desc x", "".join(rnd.choice("ACGT") for _ in range(L)), "".join(rnd.choice("@+IJ#") for _ in range(L))))
    fq = "".join(f"@{a}\n{s}\n+\n{q}\n" for a, s, q in recs).encode()
    fa = "".join(f">{a}\n" + "\n".join(s[k:k + 60] for k in range(0, len(s), 60)) + "\n" for a, s, q in recs).encode()
    half = fq.index(b"\n@r3000 ") + 1
    files = {"x.fastq": fq, "x.fasta": fa, "x.fastq.gz": gzip.compress(fq, 1),
             "multi.fastq.gz": gzip.compress(fq[:half], 1) + gzip.compress(fq[half:], 1), "b.fastq.gz": bgzf(fq),
             "b.fasta.gz": bgzf(fa), "x.fastq.bz2": bz2.compress(fq), "crlf.fastq": fq.replace(b"\n", b"\r\n"),
             "nonl.fastq": fq[:-1], "trunc.fastq.gz": gzip.compress(fq, 1)[:-100]}
    for name, data in files.items():
        (tmp_path / name).write_bytes(data)
    for name in files:
        want = "".join(f"{a}\t{s}\t{q if 'fastq' in name else ''}\n" for a, s, q in recs)
        for w, cap in ((1 << 16, None), (1 << 20, None), (1 << 30, None), (1 << 18, 777)):
            out = subprocess.run([exe, str(tmp_path / name), str(w)] + ([str(cap)] if cap else []), capture_output=True, text=True).stdout
            if name.startswith("trunc"):
                assert "#error Error while decompressing" in out, (name, w)
                continue
            body, last = out.rsplit("#windows", 1)
            assert body == want, (name, w, cap)
            assert last.split()[2] == str(len(recs)) and (w >= 1 << 30) == (last.split()[0] == "1"), (name, w, last)
        # raw windows (extract hands the text to the GPU unparsed, FASTQ and FASTA alike): whole records per window, every
        # record exactly once
        if name.startswith("trunc"):
            continue
        for w, mode in ((1 << 16, "raw"), (1 << 18, "raw"), (1 << 30, "raw")):
            out = subprocess.run([exe, str(tmp_path / name), str(w), "0", mode], capture_output=True, text=True).stdout
            body, last = out.rsplit("#windows", 1)
            assert body == want, (name, w, mode, out[-300:])
            assert last.split()[2] == str(len(recs)), (name, w, mode, last)


def test_fasta_records_longer_than_a_parse_piece(tmp_path):
    """a window is parsed in pieces on the host threads; a FASTA record longer than a piece (a chromosome) leaves
    the trailing pieces empty -- the piece that owns the record must then be the one that says where the window
    ends (plain input) and the one allowed to find its record cut off by the buffer end (gzip input)"""
    import gzip
    import numpy as np
    cli_dir = os.path.join(ROOT, "merkurio_amd/csrc/cli")
    exe = str(tmp_path / "fs")
    subprocess.run(["g++", "-std=c++17", "-O1", "-w", "-I", cli_dir, "-o", exe, os.path.join(ROOT, "tests/helpers/fastx_stream_harness.cpp"),
                    os.path.join(cli_dir, "io.cpp"), os.path.join(cli_dir, "decompress.cpp"), os.path.join(cli_dir, "util.cpp"),
                    "-lz", "-ldl", "-lpthread"], check=True)
    g = np.random.default_rng(4)
    lens = [36 << 20, 100, 40 << 20, 33 << 20, 7]
    seqs = [np.frombuffer(b"ACGT", dtype=np.uint8)[g.integers(0, 4, n)] for n in lens]

    fa = bytearray()
    for i, sq in enumerate(seqs):
        fa += f">chr{i} x\n".encode()
        body = sq.tobytes()
        if len(body) > 1000:  # 80-column lines
            lines = np.frombuffer(body[:len(body) // 80 * 80], dtype=np.uint8).reshape(-1, 80)
            fa += np.concatenate([lines, np.full((lines.shape[0], 1), 10, dtype=np.uint8)], axis=1).tobytes()
            fa += body[len(body) // 80 * 80:] + b"\n"
        else:
            fa += body + b"\n"
    (tmp_path / "g.fasta").write_bytes(bytes(fa))
    (tmp_path / "g.fasta.gz").write_bytes(gzip.compress(bytes(fa), 1))
    env = dict(os.environ, MERKURIO_IO_THREADS="8")
    ref = None
    for name in ("g.fasta", "g.fasta.gz"):
        for w in (64 << 20, 48 << 20, 1 << 30):
            out = subprocess.run([exe, str(tmp_path / name), str(w), "0", "digest"], capture_output=True, text=True, env=env).stdout
            assert "#error" not in out, (name, w, out[-300:])
            rows = [l.split("\t") for l in out.splitlines() if not l.startswith("#")]
            assert [r[0] for r in rows] == [f"chr{i} x" for i in range(len(lens))], (name, w, out[-300:])
            assert [int(r[1]) for r in rows] == lens, (name, w)
            ref = ref or [r[2] for r in rows]
            assert [r[2] for r in rows] == ref, (name, w)  # the same bytes whatever the window and the container


def test_windowed_sam_bam_reader(tmp_path, golden):
    """tag reads its input a window at a time (SamFile::open / fill): SAM text, BGZF BAM, a BAM written as ONE
    gzip member, and an uncompressed BAM must yield the same header and records whatever the window size."""
    import gzip
    import random
    import struct
    import zlib
    cli_dir = os.path.join(ROOT, "merkurio_amd/csrc/cli")
    exe = str(tmp_path / "ss")
    subprocess.run(["g++", "-std=c++17", *_HARNESS_FLAGS, "-w", "-I", cli_dir, "-o", exe, os.path.join(ROOT, "tests/helpers/sam_stream_harness.cpp"),
                    os.path.join(cli_dir, "io.cpp"), os.path.join(cli_dir, "decompress.cpp"), os.path.join(cli_dir, "util.cpp"),
                    "-lz", "-ldl", "-lpthread"], check=True)
    rnd = random.Random(9)
    n = 8000
    seqs = ["".join(rnd.choice("ACGTacgtN") for _ in range(rnd.choice((1, 36, 100, 151)))) for _ in range(n)]
    header = "@HD\tVN:1.6\tSO:unsorted\n@SQ\tSN:chr1\tLN:1000\n@CO\t" + "x" * 300 + "\n"
    sam = header + "".join(f"read{i}\t4\t*\t0\t0\t*\t*\t0\t0\t{s}\t{'I' * len(s)}" + ("\tkm:Z:OLD,VAL" if i % 97 == 0 else "") + "\tNM:i:3\n"
                           for i, s in enumerate(seqs))
    (tmp_path / "x.sam").write_text(sam)
    # the same records as BAM (unmapped, no CIGAR), built by hand
    nib = {c: v for c, v in zip("=ACMGRSVTWYHKDBN", range(16))}
    body = bytearray()
    for i, s in enumerate(seqs):
        name = f"read{i}".encode() + b"\0"
        packed = bytearray((len(s) + 1) // 2)
        for k, ch in enumerate(s.upper()):
            packed[k // 2] |= nib[ch] << (4 if k % 2 == 0 else 0)
        aux = (b"kmZOLD,VAL\0" if i % 97 == 0 else b"") + b"NMC\x03"
        rec = struct.pack("<iiBBHHHiiii", -1, -1, len(name), 0, 4680, 0, 4, len(s), -1, -1, 0) + name + bytes(packed) + bytes([40] * len(s)) + aux
        body += struct.pack("<i", len(rec)) + rec
    ht = header.encode()
    raw = b"BAM\x01" + struct.pack("<i", len(ht)) + ht + struct.pack("<i", 1) + struct.pack("<i", 5) + b"chr1\0" + struct.pack("<I", 1000) + bytes(body)

    def bgzf(data, block=0xff00):
        out = bytearray()
        for b in range(0, len(data), block):
            chunk = data[b:b + block]
            co = zlib.compressobj(6, zlib.DEFLATED, -15)
            c = co.compress(chunk) + co.flush()
            out += bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0]) + struct.pack("<H", len(c) + 25)
            out += c + struct.pack("<II", zlib.crc32(chunk), len(chunk))
        return bytes(out) + bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")

    (tmp_path / "b.bam").write_bytes(bgzf(raw))
    (tmp_path / "g.bam").write_bytes(gzip.compress(raw, 1))
    (tmp_path / "u.bam").write_bytes(raw)
    (tmp_path / "trunc.bam").write_bytes(bgzf(raw[:-7]))

    def run_h(name, w, *extra):
        return subprocess.run([exe, str(tmp_path / name), str(w), *extra], capture_output=True, text=True).stdout

    ref = {}
    for name in ("x.sam", "b.bam", "g.bam", "u.bam"):
        outs = [run_h(name, w) for w in (1 << 16, 1 << 19, 1 << 30)]
        # the next window prepared on a second thread beside the current one (SamFile::prefetch): the same windows
        assert [run_h(name, w, "prefetch") for w in (1 << 16, 1 << 19)] == outs[:2], name
        bodies = [o.rsplit("#windows", 1)[0] for o in outs]
        assert bodies[0] == bodies[1] == bodies[2], name
        assert outs[0].rsplit("#windows", 1)[1].split()[2] == str(n) and int(outs[0].rsplit("#windows", 1)[1].split()[0]) > 3, name
        assert outs[2].rsplit("#windows", 1)[1].split()[0] == "1"
        lines = bodies[0].split("\n")
        assert "".join(l + "\n" for l in lines[:3]) == header
        ref[name] = [l.split("\t|")[0] for l in lines[3:] if l]  # name, matcher's sequence, existing km value
    # what the matcher sees: upper-case sequence; the three BAM containers agree with each other and with the SAM
    want = [f"read{i}\t{s.upper()}\t" + ("1:OLD,VAL" if i % 97 == 0 else "0:") for i, s in enumerate(seqs)]
    assert ref["x.sam"] == want and ref["b.bam"] == want and ref["g.bam"] == want and ref["u.bam"] == want
    assert "#error Error during BAM record parsing: truncated file" in run_h("trunc.bam", 1 << 16)
    assert "#error Error during BAM record parsing: truncated file" in run_h("trunc.bam", 1 << 16, "prefetch")
    # a window of >= 32 MB: the record chain is walked in pieces on all host threads (starts found by four consistent records in
    # a row, every piece must end exactly on the next one's start) -- the same records as the serial walk of small windows;
    # names / qualities that look like record headers do not derail it (a false start fails the validation: serial walk)
    big = bytearray()
    rnd2 = random.Random(3)
    for i in range(180000):
        L = rnd2.choice((36, 100, 151, 250))
        name = (b"r%d" % i if i % 1000 else struct.pack("<i", 40) + b"xx") + b"\0"  # some names start like a block_size field
        s_ = bytes(rnd2.choice(b"\x11\x12\x14\x18\x21\x22\x24\x28\x41\x42\x44\x48\x81\x82\x84\x88") for _ in range((L + 1) // 2))
        rec = struct.pack("<iiBBHHHiiii", -1, -1, len(name), 0, 4680, 0, 4, L, -1, -1, 0) + name + s_ + bytes([40] * L) + b"NMC\x03"
        big += struct.pack("<i", len(rec)) + rec
    raw_big = b"BAM\x01" + struct.pack("<i", len(ht)) + ht + struct.pack("<i", 1) + struct.pack("<i", 5) + b"chr1\0" + struct.pack("<I", 1000) + bytes(big)
    assert len(raw_big) > 36 << 20
    (tmp_path / "big.bam").write_bytes(raw_big)
    o_par, o_ser = run_h("big.bam", 1 << 30), run_h("big.bam", 1 << 22)
    assert o_par.rsplit("#windows", 1)[0] == o_ser.rsplit("#windows", 1)[0] and o_par.rsplit("#windows", 1)[1].split() == ["1", "records", "180000"]
    (tmp_path / "bigtrunc.bam").write_bytes(raw_big[:-9])
    assert "#error Error during BAM record parsing: truncated file" in run_h("bigtrunc.bam", 1 << 30)
    broken = bytearray(raw_big)
    broken[len(raw_big) // 2:len(raw_big) // 2 + 4] = struct.pack("<i", 7)  # a block_size < 32 in the middle (or inside a record: then nothing happens)
    (tmp_path / "bigbroken.bam").write_bytes(bytes(broken))
    assert run_h("bigbroken.bam", 1 << 30).rsplit("#windows", 1)[0] == run_h("bigbroken.bam", 1 << 22).rsplit("#windows", 1)[0]


def test_bam_writer_pieces_queue_and_encoder(tmp_path):
    """BamWriter without a device (the --host-codec path; the device path differs in the deflate call only): records
    encoded by the allocation-free encoder, handed over in pieces of 1 / 7 / 5 000 records or one by one, cut into 65 280-byte
    members by the writer's queue and thread -- the BAM gunzips to the same stream whatever the pieces, its members are
    whole BGZF members with an end-of-file marker, and the records read back (SamFile) are the SAM lines that went in,
    optional fields of every type included."""
    import gzip
    import random
    import struct
    cli_dir = os.path.join(ROOT, "merkurio_amd/csrc/cli")
    common = [os.path.join(cli_dir, "io.cpp"), os.path.join(cli_dir, "decompress.cpp"), os.path.join(cli_dir, "util.cpp"), "-lz", "-ldl", "-lpthread"]
    wexe, rexe = str(tmp_path / "bw"), str(tmp_path / "ss")
    subprocess.run(["g++", "-std=c++17", *_HARNESS_FLAGS, "-w", "-I", cli_dir, "-o", wexe, os.path.join(ROOT, "tests/helpers/bam_writer_harness.cpp"), *common], check=True)
    subprocess.run(["g++", "-std=c++17", *_HARNESS_FLAGS, "-w", "-I", cli_dir, "-o", rexe, os.path.join(ROOT, "tests/helpers/sam_stream_harness.cpp"), *common], check=True)
    rnd = random.Random(31)
    n = 40000  # 7 MB of BAM records, > 100 members
    header = "@HD\tVN:1.6\tSO:unsorted\n@SQ\tSN:chr1\tLN:100000\n@SQ\tSN:chrUn_x\tLN:5000\n"
    lines = []
    for i in range(n):
        L = rnd.choice((0, 1, 36, 101, 150))
        seq = "".join(rnd.choice("ACGTNacgtRY=") for _ in range(L)) or "*"
        qual = "".join(chr(33 + rnd.randrange(42)) for _ in range(L)) if L and i % 11 else "*"
        mapped = i % 3 == 0 and L > 0
        cigar = f"{L}M" if mapped and i % 2 else (f"5S{L - 10}M2I3S" if mapped and L > 12 else "*")
        tags = ["NM:i:%d" % rnd.choice((0, 7, 255, 256, 70000, -1, -200, -40000)), "XA:A:q", "XF:f:1.5", "RG:Z:grp%d" % (i % 3), "XH:H:1AE301",
                "XB:B:c,-1,2,3", "XS:B:S,1,65535", "XI:B:i,-5,70000", "XE:B:f,0.5,2", "XC:B:C"][:rnd.randrange(11)]
        lines.append("\t".join([f"read{i}/x", str(rnd.choice((0, 4, 16, 99))), "chr1" if mapped else "*", str(rnd.randrange(1, 90000) if mapped else 0),
                                str(rnd.randrange(61)), cigar, rnd.choice(("=", "*", "chrUn_x")), str(rnd.randrange(5000)), str(rnd.randrange(-500, 500)),
                                seq, qual] + tags))
    (tmp_path / "in.sam").write_text(header + "\n".join(lines) + "\n")
    streams = []
    for per, run_members in ((0, 1536), (1, 2), (7, 5), (5000, 1)):  # (runs of 1 ... 5 members: dozens of runs through the writer's queue)
        out = tmp_path / f"o{per}.bam"
        r = subprocess.run([wexe, str(tmp_path / "in.sam"), str(out), str(per), str(run_members)], capture_output=True, text=True)
        assert r.stdout.strip() == f"#records {n}", (per, r.stdout, r.stderr)
        raw = out.read_bytes()
        assert raw.endswith(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
        at, members = 0, 0
        while at < len(raw):  # whole members, every one but the last two of 65 280 bytes of text
            assert raw[at:at + 4] == b"\x1f\x8b\x08\x04" and raw[at + 12:at + 16] == b"BC\x02\x00"
            at += struct.unpack_from("<H", raw, at + 16)[0] + 1
            members += 1
        assert at == len(raw) and members > 100
        streams.append(gzip.decompress(raw))
    assert streams[0] == streams[1] == streams[2] == streams[3] and streams[0][:4] == b"BAM\x01"
    back = subprocess.run([rexe, str(tmp_path / "o7.bam"), str(1 << 20)], capture_output=True, text=True).stdout
    got = [l.split("\t|", 1)[1] for l in back.split("\n") if "\t|" in l]
    # what comes back is the line that went in, up to what BAM does not keep: sequence case, 'i' values as the smallest type
    def norm(l):
        f = l.split("\t")
        f[9] = f[9].upper()
        if f[6] == "=" and f[2] == "*":
            f[6] = "*"  # "the same reference" of an unmapped read is no reference
        return "\t".join(f)
    assert len(got) == n and [norm(l) for l in got] == [norm(l) for l in lines]
    # members made elsewhere passed through (put_members, r05: `tag` hands the writer the members the device deflated): the header
    # is closed with a member of its own, the files follow as they are, in order, through the writer's queue and recycled buffers
    import zlib
    def bgzf(data, block=0xff00):
        out = bytearray()
        for b in range(0, len(data), block):
            chunk = data[b:b + block]
            co = zlib.compressobj(6, zlib.DEFLATED, -15)
            c = co.compress(chunk) + co.flush()
            out += bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0]) + struct.pack("<H", len(c) + 25) + c
            out += struct.pack("<II", zlib.crc32(chunk), len(chunk))
        return bytes(out)
    (tmp_path / "h.sam").write_text(header)
    payloads = [bytes(rnd.randrange(256) for _ in range(k)) * 50 for k in (3000, 1, 70000 // 50, 2500, 9000, 17, 4000)]
    for k, p in enumerate(payloads):
        (tmp_path / f"m{k}.bin").write_bytes(bgzf(p))
    r = subprocess.run([wexe, "--members", str(tmp_path / "h.sam"), str(tmp_path / "pass.bam"), *[str(tmp_path / f"m{k}.bin") for k in range(len(payloads))]],
                       capture_output=True, text=True)
    assert r.stdout.strip() == f"#members files {len(payloads)}", (r.stdout, r.stderr)
    raw = (tmp_path / "pass.bam").read_bytes()
    hdr_stream = streams[0][:streams[0].index(b"read0/x") - 36]  # the BAM header as the first runs wrote it (records start 36 bytes in front of the first name)
    assert gzip.decompress(raw) == hdr_stream + b"".join(payloads)
    assert raw.endswith(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))


def test_log_rows_are_formatted_like_serde_json(tmp_path):
    """the CLI writes a hit's text row and its pretty JSON object directly (several host threads format the rows of a
    batch): the bytes must be what serde_json's pretty printer + the reference's re-indentation produce
    (src/logger.rs:41-60,108-133) -- checked against Python's json module for ids with quotes, backslashes, control
    characters, tabs and non-ASCII text"""
    import json
    cli_dir = os.path.join(ROOT, "merkurio_amd/csrc/cli")
    exe = str(tmp_path / "lr")
    subprocess.run(["g++", "-std=c++17", "-O1", "-w", "-I", cli_dir, "-o", exe, os.path.join(ROOT, "tests/helpers/logrow_harness.cpp"),
                    os.path.join(cli_dir, "util.cpp"), "-lz", "-ldl", "-lpthread"], check=True)
    rows = [("reads.fastq", "r1 desc", "ACGT", 0), ("a \"quoted\" name.fq", "id\\with\\backslashes", "ACGTN", 4294967295),
            ("f.fa", "tab\there\x01\x1c ctrl", "acgt", 17), ("ünïcödé.fastq", "日本語 id", "ACGT" * 20, 18446744073709551615),
            ("f", "", "A", 7)]
    (tmp_path / "in").write_bytes("\x1e".join("\x1f".join([f, i, p, str(pos)]) for f, i, p, pos in rows).encode())
    out = subprocess.run([exe, str(tmp_path / "in")], capture_output=True, check=True).stdout
    text, js = out.split(b"\x1d")
    assert text.decode() == "".join(f"{f}\t{i}\t{p}\t{pos}\n" for f, i, p, pos in rows)
    want = []
    for f, i, p, pos in rows:
        pretty = json.dumps({"file": f, "record_id": i, "pattern": p, "position": str(pos)}, indent=2, sort_keys=True, ensure_ascii=False)
        want.append("".join("    " + line + "\n" for line in pretty.split("\n")))
    assert js.decode() == ",\n".join(want)

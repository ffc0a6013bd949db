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


def test_bz2_xz_zstd_decoders(tmp_path, golden):
    """needletail reads .gz/.bz2/.xz (zstd with the same feature); the CLI binds libbz2 / liblzma /
    libzstd at run time.  Harness around the CLI's decoder file: reference samples, multi-stream
    bzip2, multi-MB inputs, truncated inputs are errors."""
    import bz2
    import ctypes
    import lzma
    exe = str(tmp_path / "dz")
    subprocess.run(["g++", "-std=c++17", "-O1", "-DMK_DECOMPRESS_HARNESS", "-I", os.path.join(ROOT, "merkurio_amd/csrc/cli"), "-o", exe,
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

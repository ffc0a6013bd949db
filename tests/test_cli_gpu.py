"""End-to-end tests of the C++ `merkurio extract|tag` host program (GPU: it calls the HIP
library) against the reference's golden files -- the same comparisons the reference's own
fixture tests make (src/cmd_extract.rs:724-1057, src/cmd_tag.rs:696-1135), plus byte-exact
checks of the parts of the text / JSON logs that do not depend on time, version or paths."""
import json
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "merkurio_amd", "lib", "merkurio")


@pytest.fixture(scope="module", autouse=True)
def _built():
    from merkurio_amd import build, native
    build.build_all()
    if native.device_count() < 1:
        pytest.fail("no HIP device visible")


def run(args, cwd=None, check=True):
    p = subprocess.run([BIN] + args, cwd=cwd, capture_output=True)
    if check and p.returncode != 0:
        raise AssertionError(f"merkurio {' '.join(args)} -> {p.returncode}\n{p.stderr.decode()}")
    return p


def log_body(path):
    """text log without its 4 volatile header lines (src/cmd_extract.rs:746-748)"""
    return open(path, "rb").read().split(b"\n", 4)[4]


def json_stable(path):
    """(text of the matching_records array, text from the first statistics object to the end, parsed doc)"""
    t = open(path, "rb").read()
    head, rest = t.split(b'  "meta_information": ', 1)
    key = b'  "paired_end_reads_statistics": ' if b'"paired_end_reads_statistics"' in rest else b'  "pattern_hit_counts": '
    tail = key + rest.split(key, 1)[1]
    return head, tail, json.loads(t)


def check_json(got, gold):
    gh, gt, gj = json_stable(got)
    eh, et, ej = json_stable(gold)
    assert gh == eh and gt == et
    for k in ("search_algorithm", "inverted_matching", "case_insensitive", "subcommand", "program", "input_files"):
        if k == "input_files":
            assert gj["meta_information"][k]["record_file_1"] == ej["meta_information"][k]["record_file_1"]
        else:
            assert gj["meta_information"][k] == ej["meta_information"][k]
    assert list(gj["meta_information"].keys()) == list(ej["meta_information"].keys())  # same (sorted) key set


def sam_without_own_pg(path):
    return [ln for ln in open(path, "rb").read().split(b"\n") if not ln.startswith(b"@PG\tID:merkurio")]


@pytest.mark.parametrize("name,extra", [("simple", []), ("simple-inv", ["-v"])])
def test_extract_fasta_fixtures(golden, tmp_path, name, extra):
    fx = os.path.join(golden, "fixtures")
    out = tmp_path / f"{name}.extracted.fasta"
    run(["extract", "-i", os.path.join(fx, "input/simple.fasta"), "-r", "-s", "ACG", *extra, "-o", str(out),
         "-l", str(tmp_path / "x.log"), "-j", str(tmp_path / "x.json")])
    assert out.read_bytes() == open(os.path.join(fx, f"extract/{name}.extracted.fasta"), "rb").read()
    assert log_body(tmp_path / "x.log") == log_body(os.path.join(fx, f"extract/{name}.log"))
    check_json(tmp_path / "x.json", os.path.join(fx, f"extract/{name}.json"))
    head = open(tmp_path / "x.log").read().split("\n")[:4]
    assert head[0] == "#SeqKatcher extract log" and head[2].startswith("#Running merkurio version ")
    assert head[3].startswith("#Command line: ")


def test_extract_fixed_width(golden, tmp_path):
    fx = os.path.join(golden, "fixtures")
    out = tmp_path / "fw.faa"
    run(["extract", "-i", os.path.join(fx, "input/fixed-width.faa"), "-s", "DKAT", "-o", str(out), "-l", str(tmp_path / "x.log"),
         "-j", str(tmp_path / "x.json")])
    assert out.read_bytes() == open(os.path.join(fx, "extract/fixed-width.extracted.faa"), "rb").read()
    assert log_body(tmp_path / "x.log") == log_body(os.path.join(fx, "extract/fixed-width.log"))
    check_json(tmp_path / "x.json", os.path.join(fx, "extract/fixed-width.json"))


def test_extract_paired(golden, tmp_path):
    fx = os.path.join(golden, "fixtures")
    run(["extract", "-i", os.path.join(fx, "input/paired-1.fastq"), "-2", os.path.join(fx, "input/paired-2.fastq"), "-s", "CTT",
         "-o", str(tmp_path / "paired.extracted.fastq"), "-l", str(tmp_path / "x.log"), "-j", str(tmp_path / "x.json")])
    for k in (1, 2):
        assert (tmp_path / f"paired_{k}.extracted.fastq").read_bytes() == \
            open(os.path.join(fx, f"extract/paired_{k}.extracted.fastq"), "rb").read()
    assert log_body(tmp_path / "x.log") == log_body(os.path.join(fx, "extract/paired.log"))
    check_json(tmp_path / "x.json", os.path.join(fx, "extract/paired.json"))


@pytest.mark.parametrize("name,inp,extra,out", [
    ("simple", "simple.sam", ["-m"], "simple.extracted.sam"),
    ("simple-inv", "simple.sam", ["-v"], "simple-inv.extracted.sam"),
    ("simple-bam", "simple.bam", [], "simple.tagged.extracted.sam"),
    # the reference's BAM through the DEVICE inflater (by default its two members stay below the threshold and take zlib)
    ("simple-bam", "simple.bam", ["--device-codec-always"], "simple.tagged.extracted.sam"),
    ("simple-bam", "simple.bam", ["--host-codec"], "simple.tagged.extracted.sam"),
])
def test_tag_fixtures(golden, tmp_path, name, inp, extra, out):
    fx = os.path.join(golden, "fixtures")
    o = tmp_path / "out.sam"
    p = subprocess.run([BIN, "tag", "-i", os.path.join(fx, "input", inp), "-o", str(o), "-s", "CTC", "-r", "-l", str(tmp_path / "x.log"), "-j",
                        str(tmp_path / "x.json"), "-p", "2", *extra], capture_output=True, env=dict(os.environ, MERKURIO_TIMING="1"))
    assert p.returncode == 0, p.stderr.decode()
    # (timing mode says whether the device codec inflated anything)
    assert (b"BGZF inflate calls of the device codec" in p.stderr) == ("--device-codec-always" in extra)
    assert sam_without_own_pg(o) == sam_without_own_pg(os.path.join(fx, "tag", out))
    own = [ln for ln in o.read_bytes().split(b"\n") if ln.startswith(b"@PG\tID:merkurio")]
    assert len(own) == 1 and own[0].startswith(b"@PG\tID:merkurio\tPN:merkurio\tCL:") and b"\tVN:" in own[0]
    # the 5-line header of the tag log carries the tag line: body starts after 5 lines there too
    g = open(tmp_path / "x.log", "rb").read().split(b"\n", 4)
    e = open(os.path.join(fx, "tag", f"{name}.log"), "rb").read().split(b"\n", 4)
    assert g[4] == e[4] and g[0] == b"#SeqKatcher tag log"
    check_json(tmp_path / "x.json", os.path.join(fx, "tag", f"{name}.json"))


def test_tag_aho_corasick_vector(golden, tmp_path):
    """tests/fixtures/extract/log.json: tag -i simple.bam -S -s CTC AC CT AA T A C G GA AG -r -j log.json"""
    fx = os.path.join(golden, "fixtures")
    for extra in ([], ["--device-codec-always"]):
        run(["tag", "-i", os.path.join(fx, "input/simple.bam"), "-S", "-s", "CTC", "AC", "CT", "AA", "T", "A", "C", "G", "GA", "AG", "-r",
             "-j", str(tmp_path / "log.json"), *extra])
        check_json(tmp_path / "log.json", os.path.join(fx, "extract/log.json"))


def test_example_minimal_stdout(golden):
    d = os.path.join(golden, "example-minimal")
    p = run(["extract", "-f", os.path.join(d, "kmers.txt"), "-i", os.path.join(d, "sample.fasta")])
    src = open(os.path.join(d, "sample.fasta"), "rb").read()
    assert p.stdout == (src if src.endswith(b"\n") else src + b"\n")
    p = run(["tag", "-f", os.path.join(d, "kmers.txt"), "-i", os.path.join(d, "sample.sam")])
    lines = p.stdout.split(b"\n")
    assert sum(1 for ln in lines if ln and not ln.startswith(b"@")) == 3 and all(b"\tkm:Z:" in ln for ln in lines if ln and ln[:1] != b"@")


def test_example_workflow(golden, tmp_path):
    wf = os.path.join(golden, "example-workflow")
    run(["extract", "-i", os.path.join(wf, "data/mutant_R1.subset.fastq.gz"), "-2", os.path.join(wf, "data/mutant_R2.subset.fastq.gz"),
         "-f", os.path.join(wf, "significant_kmers.txt"), "-r", "-o", str(tmp_path / "mutant_extracted"), "-l",
         str(tmp_path / "x.log"), "-j", str(tmp_path / "x.json")])
    for k in (1, 2):
        assert (tmp_path / f"mutant_extracted_{k}.fastq").read_bytes() == \
            open(os.path.join(wf, f"output/mutant_extracted_{k}.fastq"), "rb").read()
    got = json.load(open(tmp_path / "x.json"))
    gold = json.load(open(os.path.join(wf, "logs/mutant_extracted.stats.json")))
    strip = lambda rows: [(r["record_id"], r["pattern"], r["position"], r["file"].split(".")[0]) for r in rows]
    assert strip(got["matching_records"]) == strip(gold["matching_records"])
    assert got["pattern_hit_counts"] == gold["pattern_hit_counts"]
    assert got["paired_end_reads_statistics"] == gold["paired_end_reads_statistics"]
    # tag golden (README.md:259): no logging
    o = tmp_path / "tagged.sam"
    run(["tag", "-i", os.path.join(wf, "output/mutant_extracted.sorted.sam"), "-f", os.path.join(wf, "significant_kmers.txt"), "-r",
         "-o", str(o)])
    assert sam_without_own_pg(o) == sam_without_own_pg(os.path.join(wf, "output/mutant_extracted.sorted.tagged.sam"))


def test_tag_bam_output_round_trip(golden, tmp_path):
    """-o x.bam: BGZF/BAM writer; read back through the BAM reader (tag x.bam -> SAM) and compare
    with the direct SAM output.  (The reference leaves BAM output untested, src/cmd_tag.rs:1134.)"""
    fx = os.path.join(golden, "fixtures/input")
    for inp in ("simple.sam", "simple.bam"):
        run(["tag", "-i", os.path.join(fx, inp), "-o", str(tmp_path / "t.bam"), "-s", "CTC", "-r"])
        run(["tag", "-i", os.path.join(fx, inp), "-o", str(tmp_path / "direct.sam"), "-s", "CTC", "-r"])
        # python can gunzip BGZF; the magic and the EOF marker must be in place
        import gzip
        raw = open(tmp_path / "t.bam", "rb").read()
        assert gzip.decompress(raw)[:4] == b"BAM\x01" and raw.endswith(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
        run(["tag", "-i", str(tmp_path / "t.bam"), "-o", str(tmp_path / "back.sam"), "-s", "ZZZZ", "-t", "zz"])
        back = [ln.rsplit(b"\tzz:Z:", 1)[0] for ln in sam_without_own_pg(tmp_path / "back.sam") if ln and not ln.startswith(b"@")]
        direct = [ln for ln in sam_without_own_pg(tmp_path / "direct.sam") if ln and not ln.startswith(b"@")]
        assert back == direct and len(direct) == 3


def _bgzf(data, block=0xff00):
    """data as BGZF (SAM spec 4.1): independent gzip members with the BC extra field + EOF marker"""
    import struct
    import zlib
    out = bytearray()
    for b in range(0, len(data), block):
        chunk = data[b:b + block]
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        c = co.compress(chunk) + co.flush()
        out += bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0]) + struct.pack("<H", len(c) + 25)
        out += c + struct.pack("<II", zlib.crc32(chunk), len(chunk))
    return bytes(out) + bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def test_bgzf_input_and_large_bam_round_trip(tmp_path):
    """many-member BGZF inputs take the parallel inflate path; BAM output of many blocks takes the
    parallel deflate path.  Results must equal those of the plain-text inputs / SAM output."""
    import gzip
    import random
    rnd = random.Random(11)
    kmers = ["".join(rnd.choice("ACGT") for _ in range(21)) for _ in range(40)]
    reads = []
    for i in range(30000):
        s = "".join(rnd.choice("ACGT") for _ in range(rnd.choice((80, 100, 151))))
        if i % 9 == 0:
            k = rnd.choice(kmers)
            o = rnd.randrange(len(s) - 21)
            s = s[:o] + k + s[o + 21:]
        reads.append(s)
    fq = "".join(f"@r{i} x\n{s}\n+\n{'I' * len(s)}\n" for i, s in enumerate(reads)).encode()
    (tmp_path / "k.txt").write_text("\n".join(kmers) + "\n")
    (tmp_path / "plain.fastq").write_bytes(fq)
    (tmp_path / "bgzf.fastq.gz").write_bytes(_bgzf(fq))
    (tmp_path / "mono.fastq.gz").write_bytes(gzip.compress(fq, 1))  # ordinary single-member gzip: serial path
    outs = {}
    for name in ("plain.fastq", "bgzf.fastq.gz", "mono.fastq.gz"):
        run(["extract", "-i", str(tmp_path / name), "-f", str(tmp_path / "k.txt"), "-o", str(tmp_path / ("o_" + name.split(".")[0])),
             "-l", str(tmp_path / (name + ".log"))])
        body = log_body(tmp_path / (name + ".log")).replace(name.encode() + b"\t", b"<file>\t")  # rows start with the file name
        outs[name] = (open(tmp_path / ("o_" + name.split(".")[0] + ".fastq"), "rb").read(), body)
    assert outs["plain.fastq"][0].count(b"\n") >= 4 * 3000
    assert outs["bgzf.fastq.gz"] == outs["plain.fastq"] == outs["mono.fastq.gz"]
    # SAM (30 k unmapped records, ~5 MB) -> tagged BAM (dozens of BGZF blocks) -> back to SAM
    sam = "@HD\tVN:1.6\tSO:unsorted\n@SQ\tSN:chr1\tLN:100000\n" + "".join(
        f"r{i}\t4\t*\t0\t0\t*\t*\t0\t0\t{s}\t{'I' * len(s)}\tNM:i:{i % 7}\n" for i, s in enumerate(reads))
    (tmp_path / "in.sam").write_text(sam)
    run(["tag", "-i", str(tmp_path / "in.sam"), "-f", str(tmp_path / "k.txt"), "-o", str(tmp_path / "direct.sam")])
    run(["tag", "-i", str(tmp_path / "in.sam"), "-f", str(tmp_path / "k.txt"), "-o", str(tmp_path / "t.bam")])
    raw = open(tmp_path / "t.bam", "rb").read()
    assert raw.count(bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0])) > 20
    assert gzip.decompress(raw)[:4] == b"BAM\x01"
    run(["tag", "-i", str(tmp_path / "t.bam"), "-s", "ZZZZ", "-t", "zz", "-o", str(tmp_path / "back.sam")])
    back = [ln.rsplit(b"\tzz:Z:", 1)[0] for ln in sam_without_own_pg(tmp_path / "back.sam") if ln and not ln.startswith(b"@")]
    direct = [ln for ln in sam_without_own_pg(tmp_path / "direct.sam") if ln and not ln.startswith(b"@")]
    assert back == direct and len(direct) == 30000
    assert sum(1 for ln in direct if b"\tkm:Z:" in ln) >= 3000
    # the members above were deflated on the device (mk_bgzf_deflate); --host-codec = zlib on the host threads, the
    # checker: the two files must gunzip to the same BAM stream (their @PG lines name the two command lines)
    run(["tag", "-i", str(tmp_path / "in.sam"), "-f", str(tmp_path / "k.txt"), "-o", str(tmp_path / "h.bam"), "--host-codec"])
    dev, host = gzip.decompress(raw), gzip.decompress(open(tmp_path / "h.bam", "rb").read())
    cut = lambda b: b.split(b"\tVN:1.0.0\n", 1)[1]
    assert cut(dev) == cut(host) and len(dev) > 5_000_000


def test_cli_errors(golden, tmp_path):
    fx = os.path.join(golden, "fixtures/input")
    assert run(["extract", "-i", os.path.join(fx, "simple.fasta")], check=False).returncode == 2  # kmers group required
    assert run(["extract", "-i", os.path.join(fx, "simple.fasta"), "-s", "A", "-f", "x"], check=False).returncode == 2
    assert run(["extract", "-i", os.path.join(fx, "simple.fasta"), "-s", "A", "-S"], check=False).returncode == 2  # -S requires logging
    assert run(["extract", "-i", os.path.join(fx, "simple.fasta"), "-s", "A", "-q", "1", "-a"], check=False).returncode == 2
    p = run(["extract", "-i", os.path.join(fx, "simple.fasta"), "-s", "A", "-l", "-j"], check=False)  # both logs to stdout
    assert p.returncode == 1 and b"Cannot use both" in p.stderr
    p = run(["extract", "-i", os.path.join(fx, "simple.fasta"), "-s", "ACG", "-l"], check=False)  # log to stdout + records to stdout
    assert p.returncode == 1 and b"Cannot write log to stdout" in p.stderr
    p = run(["extract", "-i", os.path.join(fx, "simple.fasta"), "-s", "ACG", "-q", "9", "-o", str(tmp_path / "o")], check=False)
    assert p.returncode == 1 and b"Invalid q-gram length: 9" in p.stderr
    p = run(["extract", "-i", os.path.join(fx, "paired-1.fastq"), "-2", os.path.join(fx, "simple.fasta"), "-s", "CTT"], check=False)
    assert p.returncode == 1 and b"different number of records" in p.stderr
    p = run(["tag", "-i", os.path.join(fx, "simple.sam"), "-s", "CTC", "-t", "kmx"], check=False)
    assert p.returncode == 1 and b"Tag must be exactly two characters long." in p.stderr
    p = run(["tag", "-i", os.path.join(fx, "simple.fasta"), "-s", "CTC"], check=False)
    assert p.returncode == 1 and b"Input file must be a BAM or SAM file." in p.stderr
    # -S with a log on stdout is fine, and -v is reported in the header line
    p = run(["extract", "-i", os.path.join(fx, "simple.fasta"), "-s", "ACG", "-S", "-l", "-v"])
    assert b"#Searching for 1 pattern (inverted matching)\n" in p.stdout and b"simple.fasta\tseq1\tACG\t0\n" in p.stdout


def test_compressed_inputs_bz2_xz_zstd(golden, tmp_path):
    """the reference reads .gz / .bz2 / .xz through needletail (Cargo.toml:26, helpers.rs:48-68,
    tests/data/sample.fasta.{gz,bz2,xz}); zstd rides on the same feature.  Every compression of
    the same file must extract byte-identically to the plain file, with the uncompressed type as
    the output extension (identify_uncompressed_type)."""
    import ctypes
    data = os.path.join(golden, "data")
    plain = open(os.path.join(data, "sample.fasta"), "rb").read()
    z = ctypes.CDLL("libzstd.so.1")
    z.ZSTD_compressBound.restype = ctypes.c_size_t
    z.ZSTD_compressBound.argtypes = [ctypes.c_size_t]
    z.ZSTD_compress.restype = ctypes.c_size_t
    z.ZSTD_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int]
    cap = z.ZSTD_compressBound(len(plain))
    buf = ctypes.create_string_buffer(cap)
    n_zs = z.ZSTD_compress(buf, cap, plain, len(plain), 3)
    (tmp_path / "sample.fasta.zst").write_bytes(buf.raw[:n_zs])
    outs = {}
    for name, src in (("plain", os.path.join(data, "sample.fasta")), ("gz", os.path.join(data, "sample.fasta.gz")),
                      ("bz2", os.path.join(data, "sample.fasta.bz2")), ("xz", os.path.join(data, "sample.fasta.xz")),
                      ("zst", str(tmp_path / "sample.fasta.zst"))):
        o = tmp_path / f"o_{name}"
        run(["extract", "-i", src, "-s", "CACCATGGCCAGGAGCATTCAGC", "GGCAGCGAATGAGG", "-r", "-o", str(o), "-l", str(tmp_path / f"{name}.log")])
        rows = [ln.split(b"\t", 1)[1] for ln in log_body(tmp_path / f"{name}.log").split(b"\n") if ln and not ln.startswith(b"#")]
        # identify_uncompressed_type (src/helpers.rs:59) knows gz / bz / bz2 / xz: a .zst input keeps "zst"
        outs[name] = ((tmp_path / (f"o_{name}.zst" if name == "zst" else f"o_{name}.fasta")).read_bytes(), rows)
    assert len(outs["plain"][0]) > 0 and len(outs["plain"][1]) > 0
    for name in ("gz", "bz2", "xz", "zst"):
        assert outs[name] == outs["plain"], name
    (tmp_path / "broken.fasta.bz2").write_bytes(open(os.path.join(data, "sample.fasta.bz2"), "rb").read()[:-20])
    p = run(["extract", "-i", str(tmp_path / "broken.fasta.bz2"), "-s", "ACG"], check=False)
    assert p.returncode == 1 and b"Error while decompressing" in p.stderr


def test_gpus_2_equals_gpus_1(golden, tmp_path):
    """--gpus N: contiguous record ranges on N handles (both on device 0 on a 1-GPU box), outputs in
    device order, counters summed with mk_reduce_counters (RCCL): byte-equal to --gpus 1 for the paired
    fixture and the workflow goldens, extract and tag, text + JSON logs."""
    fx = os.path.join(golden, "fixtures")
    wf = os.path.join(golden, "example-workflow")
    cases = {
        "paired": ["extract", "-i", os.path.join(fx, "input/paired-1.fastq"), "-2", os.path.join(fx, "input/paired-2.fastq"), "-s", "CTT"],
        "workflow": ["extract", "-i", os.path.join(wf, "data/mutant_R1.subset.fastq.gz"), "-2",
                     os.path.join(wf, "data/mutant_R2.subset.fastq.gz"), "-f", os.path.join(wf, "significant_kmers.txt"), "-r"],
        "single-inv": ["extract", "-i", os.path.join(fx, "input/simple.fasta"), "-r", "-s", "ACG", "-v"],
    }
    for name, base in cases.items():
        res = {}
        for g in (1, 2, 3):
            d = tmp_path / f"{name}_{g}"
            d.mkdir()
            run(base + ["-o", str(d / "out"), "-l", str(d / "x.log"), "-j", str(d / "x.json"), "--gpus", str(g), "--batch-mb", "1"])
            files = sorted(f for f in os.listdir(d) if f.startswith("out"))
            j = json.load(open(d / "x.json"))
            res[g] = ([(f, (d / f).read_bytes()) for f in files], log_body(d / "x.log"), j["matching_records"], j["pattern_hit_counts"],
                      j["summary_statistics"], j.get("paired_end_reads_statistics"))
        assert res[1] == res[2] == res[3], name
        assert len(res[1][0]) >= 1 and len(res[1][2]) > 0
    # goldens still hold with two handles
    d = tmp_path / "paired_2"
    for k in (1, 2):
        assert (d / f"out_{k}.fastq").read_bytes() == open(os.path.join(fx, f"extract/paired_{k}.extracted.fastq"), "rb").read()
    assert log_body(d / "x.log") == log_body(os.path.join(fx, "extract/paired.log"))
    # tag: workflow SAM (48 records, 24 hit) and the BAM fixture
    tag_cases = {
        "wf": ["tag", "-i", os.path.join(wf, "output/mutant_extracted.sorted.sam"), "-f", os.path.join(wf, "significant_kmers.txt"), "-r"],
        "bam-m": ["tag", "-i", os.path.join(fx, "input/simple.bam"), "-s", "CTC", "-r", "-m"],
    }
    for name, base in tag_cases.items():
        res = {}
        for g in (1, 2):
            d = tmp_path / f"tag_{name}_{g}"
            d.mkdir()
            run(base + ["-o", str(d / "o.sam"), "-l", str(d / "x.log"), "-j", str(d / "x.json"), "--gpus", str(g)])
            j = json.load(open(d / "x.json"))
            res[g] = (sam_without_own_pg(d / "o.sam"), open(d / "x.log", "rb").read().split(b"\n", 5)[5], j["matching_records"],
                      j["pattern_hit_counts"], j["summary_statistics"])
        assert res[1] == res[2], name
    assert sam_without_own_pg(tmp_path / "tag_wf_2" / "o.sam") == \
        sam_without_own_pg(os.path.join(wf, "output/mutant_extracted.sorted.tagged.sam"))


def test_tag_is_batched(tmp_path):
    """tag scans, tags and writes a slab of records at a time (--batch-mb): many small batches must
    give the same bytes as one big batch, SAM and BAM output, with and without -m"""
    import random
    rnd = random.Random(5)
    kmers = ["".join(rnd.choice("ACGT") for _ in range(31)) for _ in range(200)]
    lines = []
    for i in range(40000):
        s = "".join(rnd.choice("ACGTacgt" if i % 50 == 0 else "ACGT") for _ in range(rnd.choice((50, 150))))
        if i % 4 == 0:
            k = rnd.choice(kmers)
            o = rnd.randrange(len(s) - 31)
            s = s[:o] + k + s[o + 31:]
        extra = "\tkm:Z:OLD" if i % 1000 == 0 else ""
        lines.append(f"r{i}\t4\t*\t0\t0\t*\t*\t0\t0\t{s}\t{'I' * len(s)}{extra}\n")
    (tmp_path / "in.sam").write_text("@HD\tVN:1.6\n" + "".join(lines))
    (tmp_path / "k.txt").write_text("\n".join(kmers) + "\n")
    outs = {}
    for mb in ("1", "512"):
        for extra in ([], ["-m"]):
            o = tmp_path / f"o_{mb}_{len(extra)}.sam"
            win = ["--window-mb", "1"] if mb == "1" else []  # the small-batch runs also read the input in 1 MB windows
            run(["tag", "-i", str(tmp_path / "in.sam"), "-f", str(tmp_path / "k.txt"), "-o", str(o), "-l", str(tmp_path / "x.log"),
                 "--batch-mb", mb, *win, *extra])
            outs[(mb, len(extra))] = (sam_without_own_pg(o), open(tmp_path / "x.log", "rb").read().split(b"\n", 5)[5])
    assert outs[("1", 0)] == outs[("512", 0)] and outs[("1", 1)] == outs[("512", 1)]
    kept = [ln for ln in outs[("1", 1)][0] if ln and not ln.startswith(b"@")]
    assert 9000 < len(kept) < 40000 and all(b"\tkm:Z:" in ln for ln in kept)
    assert any(ln.endswith(b",OLD") or b",OLD," in ln or b"km:Z:OLD" in ln for ln in outs[("1", 0)][0])
    run(["tag", "-i", str(tmp_path / "in.sam"), "-f", str(tmp_path / "k.txt"), "-o", str(tmp_path / "a.bam"), "--batch-mb", "1", "--window-mb", "1"])
    run(["tag", "-i", str(tmp_path / "in.sam"), "-f", str(tmp_path / "k.txt"), "-o", str(tmp_path / "b.bam"), "--batch-mb", "512"])
    import gzip
    a, b = (gzip.decompress(open(tmp_path / f, "rb").read()).split(b"\tVN:1.0.0\n", 1)[1] for f in ("a.bam", "b.bam"))
    assert a == b and len(a) > 40000 * 100  # everything after the @PG line: reference dictionary + records
    # BAM input in windows (BGZF members inflated a group at a time), two GPUs per window: same records out
    run(["tag", "-i", str(tmp_path / "a.bam"), "-s", "ZZZZZZ", "-t", "zz", "-o", str(tmp_path / "w1.sam"), "--window-mb", "1", "--gpus", "2"])
    run(["tag", "-i", str(tmp_path / "a.bam"), "-s", "ZZZZZZ", "-t", "zz", "-o", str(tmp_path / "w2.sam")])
    assert sam_without_own_pg(tmp_path / "w1.sam") == sam_without_own_pg(tmp_path / "w2.sam")
    assert len(sam_without_own_pg(tmp_path / "w1.sam")) > 40000


def test_crlf_records_keep_their_line_ending(tmp_path):
    """needletail writes a record with the line ending it detected for that record
    (record.write(&mut writer, None), src/cmd_extract.rs:403): CRLF in, CRLF out -- FASTA (wrapped) and FASTQ"""
    fa = b">s1 first\r\nACGTACGTAC\r\nGGGTTTACGA\r\n>s2\r\nTTTTTTTTTT\r\n>s3 unix\nACGGGGTTTA\nCC\n"
    (tmp_path / "crlf.fasta").write_bytes(fa)
    run(["extract", "-i", str(tmp_path / "crlf.fasta"), "-s", "GGGTTTA", "-o", str(tmp_path / "o")])
    assert (tmp_path / "o.fasta").read_bytes() == b">s1 first\r\nACGTACGTAC\r\nGGGTTTACGA\r\n>s3 unix\nACGGGGTTTA\nCC\n"
    fq = b"@q1\r\nACGTGGGTTTA\r\n+\r\nIIIIIIIIIII\r\n@q2\r\nAAAAAAAAAAA\r\n+\r\nIIIIIIIIIII\r\n"
    (tmp_path / "crlf.fastq").write_bytes(fq)
    run(["extract", "-i", str(tmp_path / "crlf.fastq"), "-s", "GGGTTTA", "-o", str(tmp_path / "q")])
    assert (tmp_path / "q.fastq").read_bytes() == b"@q1\r\nACGTGGGTTTA\r\n+\r\nIIIIIIIIIII\r\n"


def test_extract_reads_its_input_in_windows(tmp_path):
    """extract holds one window of input at a time (--window-mb): many small windows, one big window and
    two GPUs per window must give the same bytes, single and paired, plain and gzip; a pair-count
    mismatch is reported after the common part was processed, with the reference's two messages"""
    import gzip
    import random
    rnd = random.Random(17)
    kmers = ["".join(rnd.choice("ACGT") for _ in range(25)) for _ in range(30)]
    (tmp_path / "k.txt").write_text("\n".join(kmers) + "\n")

    def reads(tag, n):
        out = []
        for i in range(n):
            s = "".join(rnd.choice("ACGT") for _ in range(rnd.choice((60, 100, 151))))
            if i % 11 == 0:
                k = rnd.choice(kmers)
                o = rnd.randrange(len(s) - 25)
                s = s[:o] + k + s[o + 25:]
            out.append(f"@p{i}/{tag}\n{s}\n+\n{'I' * len(s)}\n")
        return out

    r1, r2 = reads(1, 25000), reads(2, 25000)
    (tmp_path / "a_1.fastq").write_text("".join(r1))
    (tmp_path / "a_2.fastq").write_text("".join(r2))
    (tmp_path / "g_1.fastq.gz").write_bytes(gzip.compress("".join(r1).encode(), 1))
    (tmp_path / "g_2.fastq.gz").write_bytes(gzip.compress("".join(r2).encode(), 1))
    res = {}
    for name, f1, f2, extra in (("big", "a_1.fastq", "a_2.fastq", ["--window-mb", "1024"]),
                                ("small", "a_1.fastq", "a_2.fastq", ["--window-mb", "1", "--batch-mb", "1"]),
                                ("small-gz", "g_1.fastq.gz", "g_2.fastq.gz", ["--window-mb", "1"]),
                                ("small-2gpu", "a_1.fastq", "a_2.fastq", ["--window-mb", "1", "--gpus", "2"])):
        d = tmp_path / name
        d.mkdir()
        run(["extract", "-i", str(tmp_path / f1), "-2", str(tmp_path / f2), "-f", str(tmp_path / "k.txt"), "-r", "-o", str(d / "o"),
             "-l", str(d / "x.log"), *extra])
        body = log_body(d / "x.log").replace(b"g_1.fastq.gz\t", b"a_1.fastq\t").replace(b"g_2.fastq.gz\t", b"a_2.fastq\t")
        res[name] = ((d / "o_1.fastq").read_bytes(), (d / "o_2.fastq").read_bytes(), body)
    assert res["big"][0].count(b"\n") >= 4 * 2000
    assert res["big"] == res["small"] == res["small-gz"] == res["small-2gpu"]
    run(["extract", "-i", str(tmp_path / "a_1.fastq"), "-f", str(tmp_path / "k.txt"), "-o", str(tmp_path / "s_big")])
    run(["extract", "-i", str(tmp_path / "g_1.fastq.gz"), "-f", str(tmp_path / "k.txt"), "-o", str(tmp_path / "s_small"), "--window-mb", "1"])
    assert (tmp_path / "s_big.fastq").read_bytes() == (tmp_path / "s_small.fastq").read_bytes() != b""
    # pair-count mismatches surface where the reference notices them: at the end of the shorter file
    (tmp_path / "short_2.fastq").write_text("".join(r2[:-1]))
    p = run(["extract", "-i", str(tmp_path / "a_1.fastq"), "-2", str(tmp_path / "short_2.fastq"), "-f", str(tmp_path / "k.txt"),
             "-o", str(tmp_path / "m"), "--window-mb", "1"], check=False)
    assert p.returncode == 1 and b"Error during FASTQ record parsing of second file" in p.stderr
    (tmp_path / "short_1.fastq").write_text("".join(r1[:-3]))
    p = run(["extract", "-i", str(tmp_path / "short_1.fastq"), "-2", str(tmp_path / "a_2.fastq"), "-f", str(tmp_path / "k.txt"),
             "-o", str(tmp_path / "m2"), "--window-mb", "1"], check=False)
    assert p.returncode == 1 and b"different number of records" in p.stderr


def test_device_ingest_equals_host_ingest(tmp_path):
    """extract on a single FASTQ: the default path uploads the window's text and indexes it on the GPU
    (mk_extract_fastq_text); --host-ingest parses on the host threads, the byte-identical checker of the device's index.
    Same kept records, text log and JSON log for: plain / gzip / BGZF-less multi-window input, '@' and '+' opening
    quality lines, '+id' third lines, CRLF, no final newline, trimmed reads; a file with a blank line in the middle
    (the device refuses that window, the host reader takes over from there) and a malformed record (the reference's
    parse error, after the good windows were written)."""
    import gzip
    import random
    rnd = random.Random(12)
    kmers = ["".join(rnd.choice("ACGT") for _ in range(31)) for _ in range(50)]
    (tmp_path / "k.txt").write_text("\n".join(kmers) + "\n")

    def reads(n, lens=(150,), eol="\n", plus_id=False):
        out = []
        for i in range(n):
            L = rnd.choice(lens)
            s = [rnd.choice("ACGT") for _ in range(L)]
            if L >= 31 and i % 4 == 0:
                k = rnd.choice(kmers)
                o = rnd.randrange(0, L - 30)
                s[o:o + 31] = k
            q = "".join(rnd.choice("@+IJ#5A") for _ in range(L))
            out.append(f"@r{i} d/{i}{eol}{''.join(s)}{eol}+{('r%d' % i) if plus_id and i % 2 else ''}{eol}{q}{eol}")
        return out

    plain = reads(6000)
    files = {
        "plain.fastq": "".join(plain),
        "crlf.fastq": "".join(reads(3000, eol="\r\n", plus_id=True)),
        "ragged_nonl.fastq": "".join(reads(4000, lens=(36, 75, 150, 151, 0)))[:-1],
        "blank_line.fastq": "".join(plain[:2500]) + "\n" + "".join(plain[2500:]),
    }
    for name, text in files.items():
        (tmp_path / name).write_bytes(text.encode())
    (tmp_path / "plain.fastq.gz").write_bytes(gzip.compress(files["plain.fastq"].encode(), 1))
    # bgzip'ed (r04): the default path does not inflate these on the host at all -- windows of members go to
    # mk_extract_fastq_bgzf, the unfinished record of a window is the next one's head (members of 20 000 bytes: every window
    # ends inside a record); the blank line sends the host reader to the place in the member chain where its window began
    bz = {n[:-6] + ".bgzf.fastq.gz": n for n in files}  # (the output's extension is derived from the input's name)
    for name, src in bz.items():
        (tmp_path / name).write_bytes(_bgzf(files[src].encode(), 20000))
    for name in list(files) + ["plain.fastq.gz"] + list(bz):
        res = {}
        for mode in ("device", "host"):
            d = tmp_path / f"{name}.{mode}"
            d.mkdir()
            extra = ["--host-ingest"] if mode == "host" else []
            run(["extract", "-i", str(tmp_path / name), "-f", str(tmp_path / "k.txt"), "-o", str(d / "out"), "-l", str(d / "log.txt"), "-j",
                 str(d / "log.json"), "--window-mb", "1"] + extra)
            body = log_body(d / "log.txt").replace(name.encode() + b"\t", b"<file>\t")
            res[mode] = (open(d / "out.fastq", "rb").read(), body, json_stable(d / "log.json")[:2])
        assert res["device"] == res["host"], name
        assert res["device"][0].count(b"\n@r") > 500
        if name in bz:  # ... and the same records and rows as the plain file
            plain_out = tmp_path / f"{bz[name]}.device"
            assert res["device"][0] == open(plain_out / "out.fastq", "rb").read(), name
            assert res["device"][1] == log_body(plain_out / "log.txt").replace(bz[name].encode() + b"\t", b"<file>\t"), name
    # -v with a log: the rows name records that are not kept -- the device path brings the whole text of a window back then
    res = []
    for extra in ([], ["--host-ingest"]):
        d = tmp_path / ("inv" + "".join(extra))
        d.mkdir()
        run(["extract", "-i", str(tmp_path / "plain.bgzf.fastq.gz"), "-f", str(tmp_path / "k.txt"), "-v", "-o", str(d / "out"), "-l", str(d / "log.txt"),
             "--window-mb", "1"] + extra)
        res.append((open(d / "out.fastq", "rb").read(), log_body(d / "log.txt")))
    assert res[0] == res[1] and res[0][0].count(b"\n@r") > 3000
    # a malformed record in the third window: both paths write the records before it and end with the reference's message
    bad = "".join(plain[:5000]) + "@broken\nACGT\n+\nII\n" + "".join(plain[5000:])
    (tmp_path / "bad.fastq").write_text(bad)
    outs = []
    for extra in ([], ["--host-ingest"]):
        p = run(["extract", "-i", str(tmp_path / "bad.fastq"), "-f", str(tmp_path / "k.txt"), "--window-mb", "1"] + extra, check=False)
        assert p.returncode != 0 and b"Error during FASTQ/A record parsing." in p.stderr
        outs.append(p.stdout)
    assert outs[0] == outs[1] and outs[0].count(b"\n@r") > 300
    (tmp_path / "bad.bgzf.fastq.gz").write_bytes(_bgzf(bad.encode(), 20000))
    p = run(["extract", "-i", str(tmp_path / "bad.bgzf.fastq.gz"), "-f", str(tmp_path / "k.txt"), "--window-mb", "1"], check=False)
    # (its windows are cut at member boundaries: the records written before the failing window are a prefix of the same output)
    assert p.returncode != 0 and b"Error during FASTQ/A record parsing." in p.stderr
    assert (outs[0].startswith(p.stdout) or p.stdout.startswith(outs[0])) and p.stdout.count(b"\n@r") > 300
    # a damaged member: the reference's reader would fail on it as well; no record of that window is written
    blob = bytearray(_bgzf(files["plain.fastq"].encode(), 20000))
    blob[len(blob) // 2] ^= 0x20
    (tmp_path / "damaged.bgzf.fastq.gz").write_bytes(bytes(blob))
    for extra in ([], ["--host-ingest"], ["--host-codec"]):
        p = run(["extract", "-i", str(tmp_path / "damaged.bgzf.fastq.gz"), "-f", str(tmp_path / "k.txt"), "--window-mb", "1"] + extra, check=False)
        assert p.returncode != 0 and b"Error while decompressing" in p.stderr, extra


def test_bgzip_inputs_the_device_path_hands_back(golden, tmp_path):
    """a bgzip'ed input goes to mk_extract_fastq_bgzf first; what is not plain FASTQ comes back to the host reader at member 0:
    FASTA (the reference's sample), wrapped FASTQ; and the degenerate files: only an end-of-file member, one read"""
    d = os.path.join(golden, "example-minimal")
    fasta = open(os.path.join(d, "sample.fasta"), "rb").read()
    (tmp_path / "s.fasta.gz").write_bytes(_bgzf(fasta * 40, 3000))
    (tmp_path / "s.fasta").write_bytes(fasta * 40)
    outs = [run(["extract", "-f", os.path.join(d, "kmers.txt"), "-i", str(tmp_path / n)]).stdout for n in ("s.fasta.gz", "s.fasta")]
    assert outs[0] == outs[1] and outs[0].count(b">") >= 40
    wrapped = b"".join(b"@w%d\nACGTACGTAC\nGTACGTTTTT\n+\nIIIIIIIIII\nIIIIIIIIII\n" % i for i in range(500))
    (tmp_path / "w.fastq.gz").write_bytes(_bgzf(wrapped, 5000))
    (tmp_path / "w.fastq").write_bytes(wrapped)
    outs = [run(["extract", "-s", "ACGTTT", "-i", str(tmp_path / n), "--window-mb", "1"], check=False) for n in ("w.fastq.gz", "w.fastq")]
    assert (outs[0].returncode, outs[0].stdout, outs[0].stderr) == (outs[1].returncode, outs[1].stdout, outs[1].stderr)
    (tmp_path / "e.fastq.gz").write_bytes(_bgzf(b""))
    p = run(["extract", "-s", "ACGT", "-i", str(tmp_path / "e.fastq.gz")], check=False)
    (tmp_path / "e.fastq").write_bytes(b"")
    q = run(["extract", "-s", "ACGT", "-i", str(tmp_path / "e.fastq")], check=False)
    assert (p.returncode, p.stdout) == (q.returncode, q.stdout)
    (tmp_path / "one.fastq.gz").write_bytes(_bgzf(b"@a\nTTACGTTT\n+\nIIIIIIII"))
    assert run(["extract", "-s", "ACGT", "-i", str(tmp_path / "one.fastq.gz")]).stdout == b"@a\nTTACGTTT\n+\nIIIIIIII\n"


def test_windows_of_pairs_fasta_and_several_gpus_equal_the_host_reader(golden, tmp_path):
    """r05: BASELINE's own shapes on the window path (extract_windows.cpp -> mk_extract_window) against --host-ingest, the host
    reader: paired FASTQ (src/cmd_extract.rs:412-418,463-612) plain / gzip / bgzip'ed with mates of different record sizes (the
    windows of the two files hold different numbers of records: leftovers carried), CRLF and trimmed mates; FASTA
    (src/cmd_extract.rs:281-282) wrapped, CRLF, one-line, bgzip'ed, with a record larger than a window; each also dealt to
    --gpus 2 / 3; pair-count mismatches with the reference's two messages after the common part has been written."""
    import gzip
    import random
    rnd = random.Random(77)
    kmers = ["".join(rnd.choice("ACGT") for _ in range(31)) for _ in range(40)]
    (tmp_path / "k.txt").write_text("\n".join(kmers) + "\n")

    def seq(L, i):
        s = [rnd.choice("ACGT") for _ in range(L)]
        if L >= 31 and i % 5 == 0:
            o = rnd.randrange(0, L - 30)
            s[o:o + 31] = rnd.choice(kmers)
        return "".join(s)

    def mate(n, tag, lens, eol="\n", id_pad=""):
        return [f"@p{i}{id_pad}/{tag}{eol}{(s := seq(rnd.choice(lens), i))}{eol}+{eol}{'I' * len(s)}{eol}" for i in range(n)]

    def results(d, paired, ext):
        outs = [open(d / (f"out_{k}.{ext}" if paired else f"out.{ext}"), "rb").read() for k in ((1, 2) if paired else (0,))]
        return outs, log_body(d / "log.txt"), json_stable(d / "log.json")[:2]

    def both_ways(tag, args, paired, ext, extra_sets=((), ("--gpus", "2"), ("--gpus", "3"))):
        res = {}
        for mode in ("host",) + tuple("dev" + "".join(x) for x in extra_sets):
            d = tmp_path / f"{tag}.{mode}"
            d.mkdir()
            extra = ["--host-ingest"] if mode == "host" else list(extra_sets[[("dev" + "".join(x)) for x in extra_sets].index(mode)])
            run(["extract", *args, "-f", str(tmp_path / "k.txt"), "-o", str(d / "out"), "-l", str(d / "log.txt"), "-j", str(d / "log.json"),
                 "--window-mb", "1", *extra])
            res[mode] = results(d, paired, ext)
        for mode in res:
            assert res[mode] == res["host"], (tag, mode)
        return res["host"]

    # ---- pairs: mate 1 with long ids (fewer records per 1 MB window than mate 2), trimmed mate 2
    r1, r2 = mate(9000, 1, (150,), id_pad=" a longer description than its mate carries"), mate(9000, 2, (50, 100, 150))
    (tmp_path / "a_1.fastq").write_text("".join(r1))
    (tmp_path / "a_2.fastq").write_text("".join(r2))
    base = both_ways("pair", ["-i", str(tmp_path / "a_1.fastq"), "-2", str(tmp_path / "a_2.fastq")], True, "fastq")
    assert base[0][0].count(b"\n@p") > 1500
    (tmp_path / "g_1.fastq.gz").write_bytes(gzip.compress("".join(r1).encode(), 1))
    (tmp_path / "g_2.fastq.gz").write_bytes(gzip.compress("".join(r2).encode(), 1))
    gz = both_ways("pair-gz", ["-i", str(tmp_path / "g_1.fastq.gz"), "-2", str(tmp_path / "g_2.fastq.gz")], True, "fastq", ((), ("--gpus", "2")))
    assert gz[0] == base[0]
    (tmp_path / "b_1.fastq.gz").write_bytes(_bgzf("".join(r1).encode(), 20000))
    (tmp_path / "b_2.fastq.gz").write_bytes(_bgzf("".join(r2).encode(), 65280))
    bz = both_ways("pair-bgzf", ["-i", str(tmp_path / "b_1.fastq.gz"), "-2", str(tmp_path / "b_2.fastq.gz")], True, "fastq", ((), ("--gpus", "2")))
    assert bz[0] == base[0]
    # one plain, one bgzip'ed; -v; no logging at all
    d = tmp_path / "mixed"
    d.mkdir()
    run(["extract", "-i", str(tmp_path / "a_1.fastq"), "-2", str(tmp_path / "b_2.fastq.gz"), "-f", str(tmp_path / "k.txt"), "-o", str(d / "out"), "--window-mb", "1"])
    assert [open(d / f"out_{k}.fastq", "rb").read() for k in (1, 2)] == base[0]
    for extra in ([], ["--host-ingest"]):
        dd = tmp_path / ("inv" + "".join(extra))
        dd.mkdir()
        run(["extract", "-i", str(tmp_path / "b_1.fastq.gz"), "-2", str(tmp_path / "b_2.fastq.gz"), "-f", str(tmp_path / "k.txt"), "-v", "-o", str(dd / "out"),
             "-l", str(dd / "log.txt"), "--window-mb", "1", *extra])
    assert [open(tmp_path / "inv" / f"out_{k}.fastq", "rb").read() for k in (1, 2)] == \
        [open(tmp_path / "inv--host-ingest" / f"out_{k}.fastq", "rb").read() for k in (1, 2)]
    assert log_body(tmp_path / "inv" / "log.txt") == log_body(tmp_path / "inv--host-ingest" / "log.txt")
    # CRLF mates
    c1, c2 = mate(3000, 1, (150,), eol="\r\n"), mate(3000, 2, (75, 150), eol="\r\n")
    (tmp_path / "c_1.fastq").write_text("".join(c1), newline="")
    (tmp_path / "c_2.fastq").write_text("".join(c2), newline="")
    both_ways("pair-crlf", ["-i", str(tmp_path / "c_1.fastq"), "-2", str(tmp_path / "c_2.fastq")], True, "fastq", ((),))
    # pair-count mismatches: the common part is written, then the reference's message (both file orders, plain and bgzip'ed)
    (tmp_path / "short_2.fastq").write_text("".join(r2[:-2]))
    (tmp_path / "short_2.fastq.gz").write_bytes(_bgzf("".join(r2[:-2]).encode(), 30000))
    for second in ("short_2.fastq", "short_2.fastq.gz"):
        outs = []
        for extra in ([], ["--host-ingest"], ["--gpus", "2"]):
            dd = tmp_path / ("mm" + second + "".join(extra))
            dd.mkdir()
            p = run(["extract", "-i", str(tmp_path / "a_1.fastq"), "-2", str(tmp_path / second), "-f", str(tmp_path / "k.txt"), "-o", str(dd / "o"),
                     "--window-mb", "1", *extra], check=False)
            assert p.returncode == 1 and b"Error during FASTQ record parsing of second file" in p.stderr, (second, extra)
            outs.append([open(dd / f"o_{k}.fastq", "rb").read() for k in (1, 2)])
        assert outs[0] == outs[1] == outs[2] and outs[0][0].count(b"\n@p") > 1400
    (tmp_path / "short_1.fastq").write_text("".join(r1[:-3]))
    for extra in ([], ["--host-ingest"]):
        p = run(["extract", "-i", str(tmp_path / "short_1.fastq"), "-2", str(tmp_path / "a_2.fastq"), "-f", str(tmp_path / "k.txt"), "-o", str(tmp_path / "m2"),
                 "--window-mb", "1", *extra], check=False)
        assert p.returncode == 1 and b"different number of records" in p.stderr

    # ---- FASTA: wrapped at 70, CRLF, one line per sequence, a 3 Mbp record (three windows of 1 MB), empty sequences
    def fasta(n, width, eol="\n"):
        out = []
        for i in range(n):
            s = seq(rnd.choice((0, 40, 69, 70, 71, 500, 5000)), i)
            lines = [s[k:k + width] for k in range(0, len(s), width)] if width else [s]
            out.append(f">c{i} len={len(s)}{eol}" + "".join(ln + eol for ln in lines))
        return "".join(out)

    big = ">big one\n" + "\n".join("".join(rnd.choice("ACGT") for _ in range(60)) for _ in range(50000)) + "\n" + kmers[3][:20] + "\n" + kmers[3][20:] + "\n"
    texts = {"w70.fasta": fasta(2500, 70), "crlf.fasta": fasta(800, 60, "\r\n"), "oneline.fa": fasta(1500, 0) + big + fasta(200, 70)}
    for name, text in texts.items():
        (tmp_path / name).write_text(text, newline="")
        ext = name.rsplit(".", 1)[1]
        r = both_ways(name, ["-i", str(tmp_path / name)], False, ext, ((), ("--gpus", "2")))
        assert r[0][0].count(b">c") > 100
        (tmp_path / (name + ".gz")).write_bytes(_bgzf(text.encode(), 30000))
        rz = both_ways(name + ".bgzf", ["-i", str(tmp_path / (name + ".gz"))], False, ext, ((),))
        assert rz[0] == r[0]
    assert b">big one" in both_ways("big", ["-i", str(tmp_path / "oneline.fa"), "-r"], False, "fa", ((),))[0][0]
    # the reference's own FASTA fixtures with 1 MB windows and two devices
    fx = os.path.join(golden, "fixtures")
    for name, kmer, flags, gold in (("simple.fasta", "ACG", ["-r"], "simple"), ("fixed-width.faa", "DKAT", [], "fixed-width")):
        dd = tmp_path / ("fx" + gold)
        dd.mkdir()
        ext = name.rsplit(".", 1)[1]
        run(["extract", "-i", os.path.join(fx, "input", name), "-s", kmer, *flags, "-o", str(dd / "out"), "-l", str(dd / "x.log"), "-j", str(dd / "x.json"),
             "--gpus", "2", "--window-mb", "1"])
        assert open(dd / f"out.{ext}", "rb").read() == open(os.path.join(fx, "extract", f"{gold}.extracted.{ext}"), "rb").read()
        assert log_body(dd / "x.log") == log_body(os.path.join(fx, "extract", f"{gold}.log"))
        check_json(dd / "x.json", os.path.join(fx, "extract", f"{gold}.json"))


def _bam_stream(path):
    """the inflated BAM stream of a file, from its first record on (behind the header text, whose @PG line names the command line)"""
    import gzip
    import struct
    raw = gzip.decompress(open(path, "rb").read())
    assert raw[:4] == b"BAM\x01"
    p = 8 + struct.unpack_from("<i", raw, 4)[0]
    n_ref = struct.unpack_from("<i", raw, p)[0]
    p += 4
    for _ in range(n_ref):
        p += 8 + struct.unpack_from("<i", raw, p)[0]
    return raw[p:]


def test_tag_bam_records_resident_on_the_device(tmp_path):
    """BAM -> BAM keeps the records on the device between inflate and deflate (mk_tag_bam_window, cli/tag_windows.cpp); --host-ingest is
    the r04 path (host record index, host tag append): the two files must hold the same BAM stream and the same logs -- with and
    without -m / -v / -S, in small windows with the unfinished record carried over, and where a window is left to the host reader
    (a kept record that already has the tag; a truncated file, whose error the host reader words)."""
    import random
    rnd = random.Random(3)
    kmers = ["".join(rnd.choice("ACGT") for _ in range(25)) for _ in range(60)]
    (tmp_path / "k.txt").write_text("\n".join(kmers) + "\n")
    lines = []
    for i in range(40000):
        s = "".join(rnd.choice("ACGT") for _ in range(rnd.choice((90, 100, 151, 33))))
        if i % 7 == 0:
            k = rnd.choice(kmers)
            o = rnd.randrange(len(s) - 25) if len(s) > 25 else 0
            s = (s[:o] + k + s[o + 25:])[:max(len(s), 25)]
        extra = "\tNM:i:%d\tRG:Z:g%d" % (i % 5, i % 3) if i % 2 else ""
        lines.append(f"q{i}_{rnd.randrange(10**6)}\t0\tchr1\t{i + 1}\t60\t{len(s)}M\t*\t0\t0\t{s}\t{'F' * len(s)}{extra}\n")
    sam = "@HD\tVN:1.6\tSO:unsorted\n@SQ\tSN:chr1\tLN:1000000\n" + "".join(lines)
    (tmp_path / "in.sam").write_text(sam)
    run(["tag", "-i", str(tmp_path / "in.sam"), "-s", "ZZZZZ", "-t", "zz", "-o", str(tmp_path / "in.bam")])  # a BAM of ~8 MB of records
    for name, extra in (("all", []), ("m", ["-m"]), ("v", ["-v"])):
        for window in ([], ["--window-mb", "1"]):
            tagd = "w" if window else "d"
            p = subprocess.run([BIN, "tag", "-i", str(tmp_path / "in.bam"), "-f", str(tmp_path / "k.txt"), "-o", str(tmp_path / f"{name}_{tagd}.bam"), "-l",
                                str(tmp_path / f"{name}_{tagd}.log"), "-j", str(tmp_path / f"{name}_{tagd}.json"), *extra, *window], capture_output=True,
                               env=dict(os.environ, MERKURIO_TIMING="1"))
            assert p.returncode == 0, p.stderr.decode()
            import re
            done = re.search(rb"\[timing\] (\d+) of (\d+) windows on the device \((\d+) in flight\)", p.stderr)
            assert done and done.group(1) == done.group(2) and b"left to the host reader" not in p.stderr
            if window:
                assert int(done.group(1)) >= 5 and int(done.group(3)) == 2
        run(["tag", "-i", str(tmp_path / "in.bam"), "-f", str(tmp_path / "k.txt"), "-o", str(tmp_path / f"{name}_h.bam"), "-l", str(tmp_path / f"{name}_h.log"),
             "-j", str(tmp_path / f"{name}_h.json"), "--host-ingest", *extra])
        host = _bam_stream(tmp_path / f"{name}_h.bam")
        assert len(host) > (100_000 if name == "m" else 4_000_000)
        for tagd in ("d", "w"):
            assert _bam_stream(tmp_path / f"{name}_{tagd}.bam") == host
            assert log_body(tmp_path / f"{name}_{tagd}.log") == log_body(tmp_path / f"{name}_h.log")
            check_json(tmp_path / f"{name}_{tagd}.json", tmp_path / f"{name}_h.json")
    # --gpus 2 (on a one-GPU box both handles' devices are the same card: four windows in flight): windows dealt to the devices in turn,
    # counters per device summed at the end -- the same stream, the same logs
    p = subprocess.run([BIN, "tag", "-i", str(tmp_path / "in.bam"), "-f", str(tmp_path / "k.txt"), "-o", str(tmp_path / "g2.bam"), "-l", str(tmp_path / "g2.log"),
                        "-j", str(tmp_path / "g2.json"), "--gpus", "2", "--window-mb", "1"], capture_output=True, env=dict(os.environ, MERKURIO_TIMING="1"))
    assert p.returncode == 0, p.stderr.decode()
    assert re.search(rb"windows on the device \(4 in flight\)", p.stderr) and b"left to the host reader" not in p.stderr
    assert _bam_stream(tmp_path / "g2.bam") == _bam_stream(tmp_path / "all_h.bam")
    assert log_body(tmp_path / "g2.log") == log_body(tmp_path / "all_h.log")
    check_json(tmp_path / "g2.json", tmp_path / "all_h.json")
    # -S: no output, logs only
    run(["tag", "-i", str(tmp_path / "in.bam"), "-f", str(tmp_path / "k.txt"), "-S", "-l", str(tmp_path / "s_d.log")])
    assert log_body(tmp_path / "s_d.log") == log_body(tmp_path / "all_h.log")
    # tagging the tagged file again with the same tag: the kept records carry it already -> the reference's merge rule (the found
    # patterns and the items of the existing value, sorted, unique), on the device too
    for extra in ([], ["-m"], ["-m", "--window-mb", "1"]):
        p = subprocess.run([BIN, "tag", "-i", str(tmp_path / "all_d.bam"), "-s", kmers[0], kmers[1], kmers[2][:-1] + "A", "-o", str(tmp_path / "again_d.bam"), *extra],
                           capture_output=True, env=dict(os.environ, MERKURIO_TIMING="1"))
        assert p.returncode == 0 and b"left to the host reader" not in p.stderr and b"windows on the device" in p.stderr
        run(["tag", "-i", str(tmp_path / "all_d.bam"), "-s", kmers[0], kmers[1], kmers[2][:-1] + "A", "-o", str(tmp_path / "again_h.bam"), "--host-ingest",
             *[e for e in extra if e in ("-m",)]])
        assert _bam_stream(tmp_path / "again_d.bam") == _bam_stream(tmp_path / "again_h.bam")
    # a field of the tag's name that is not a string: the window is handed to the host reader, which words the reference's refusal --
    # in small windows after the windows in front of it have been written on the device
    import gzip
    raw_in = gzip.decompress(open(tmp_path / "in.bam", "rb").read())
    cut_at = len(raw_in) - len(_bam_stream(tmp_path / "in.bam"))
    recs_bytes = raw_in[cut_at:]
    first_len = int.from_bytes(recs_bytes[:4], "little")
    odd = bytearray(recs_bytes[:4 + first_len]) + b"kmi" + (7).to_bytes(4, "little")
    odd[0:4] = (first_len + 7).to_bytes(4, "little")
    (tmp_path / "odd.bam").write_bytes(_bgzf(raw_in + bytes(odd)))
    for extra in ([], ["--window-mb", "1"], ["--host-ingest"]):
        p = subprocess.run([BIN, "tag", "-i", str(tmp_path / "odd.bam"), "-f", str(tmp_path / "k.txt"), "-o", str(tmp_path / "odd_out.bam"), *extra],
                           capture_output=True, env=dict(os.environ, MERKURIO_TIMING="1"))
        assert p.returncode == 1 and b"Invalid tag value format. Expected string value." in p.stderr
        if extra == ["--window-mb", "1"]:
            assert b"left to the host reader (existing tag)" in p.stderr
    # a BAM without records, and one whose records are longer than a window (1 MB windows, records of 2.5 MB: a window then holds no
    # whole record and its text is carried on as the next window's head)
    import struct
    hdr_only = raw_in[:cut_at]
    (tmp_path / "empty.bam").write_bytes(_bgzf(hdr_only))
    def long_record(name, n_bases, seed):
        r2 = random.Random(seed)
        seq = bytes(r2.choice(b"ACGT") for _ in range(n_bases))
        packed = bytearray((n_bases + 1) // 2)
        for k, ch in enumerate(seq):
            packed[k >> 1] |= b"=ACMGRSVTWYHKDBN".index(ch) << (4 if k % 2 == 0 else 0)
        body = struct.pack("<iiBBHHHiiii", 0, 5, len(name) + 1, 60, 4680, 0, 4, n_bases, -1, -1, 0) + name + b"\0" + bytes(packed) + bytes([20]) * n_bases
        return struct.pack("<i", len(body)) + body, seq
    big = []
    for k in range(4):
        rec, seq = long_record(b"long%d" % k, 1_700_000, k)
        big.append(rec)
    (tmp_path / "big.bam").write_bytes(_bgzf(hdr_only + b"".join(big)))
    for name in ("empty", "big"):
        for extra, tagd in (([], "d"), (["--window-mb", "1"], "w"), (["--host-ingest"], "h")):
            p = subprocess.run([BIN, "tag", "-i", str(tmp_path / f"{name}.bam"), "-f", str(tmp_path / "k.txt"), "-o", str(tmp_path / f"{name}_{tagd}.bam"), *extra],
                               capture_output=True, env=dict(os.environ, MERKURIO_TIMING="1"))
            assert p.returncode == 0, p.stderr.decode()
            if tagd != "h":
                assert b"left to the host reader" not in p.stderr
        assert _bam_stream(tmp_path / f"{name}_d.bam") == _bam_stream(tmp_path / f"{name}_h.bam") == _bam_stream(tmp_path / f"{name}_w.bam")
    assert len(_bam_stream(tmp_path / "empty_d.bam")) == 0 and len(_bam_stream(tmp_path / "big_d.bam")) > 4 * 2_500_000
    # a truncated file: both paths end with the host reader's message
    raw = open(tmp_path / "in.bam", "rb").read()
    text = gzip.decompress(raw)
    (tmp_path / "cut.bam").write_bytes(_bgzf(text[:len(text) - 11]))
    for extra in ([], ["--host-ingest"]):
        p = run(["tag", "-i", str(tmp_path / "cut.bam"), "-f", str(tmp_path / "k.txt"), "-o", str(tmp_path / "cut_out.bam"), *extra], check=False)
        assert p.returncode == 1 and b"truncated file" in p.stderr

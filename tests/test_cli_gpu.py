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
])
def test_tag_fixtures(golden, tmp_path, name, inp, extra, out):
    fx = os.path.join(golden, "fixtures")
    o = tmp_path / "out.sam"
    run(["tag", "-i", os.path.join(fx, "input", inp), "-o", str(o), "-s", "CTC", "-r", "-l", str(tmp_path / "x.log"), "-j",
         str(tmp_path / "x.json"), "-p", "2", *extra])
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
    run(["tag", "-i", os.path.join(fx, "input/simple.bam"), "-S", "-s", "CTC", "AC", "CT", "AA", "T", "A", "C", "G", "GA", "AG", "-r",
         "-j", str(tmp_path / "log.json")])
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

"""Tiny FASTA/FASTQ/SAM readers used by the tests to turn golden files into record lists.
Test infrastructure; follows needletail's observable behaviour as listed in SURVEY.md §5
(id = header line minus the marker, seq = lines joined without newlines)."""
import gzip


def _open(path):
    with open(path, "rb") as f:
        magic = f.read(2)
    return gzip.open(path, "rb") if magic == b"\x1f\x8b" else open(path, "rb")


def read_fastx(path):
    """-> list of (id: bytes, seq: bytes)"""
    with _open(path) as f:
        data = f.read()
    lines = data.split(b"\n")
    recs = []
    i = 0
    while i < len(lines):
        ln = lines[i].rstrip(b"\r")
        if not ln:
            i += 1
            continue
        if ln[:1] == b">":
            rid = ln[1:]
            i += 1
            seq = []
            while i < len(lines) and lines[i][:1] != b">":
                seq.append(lines[i].rstrip(b"\r"))
                i += 1
            recs.append((rid, b"".join(seq)))
        elif ln[:1] == b"@":
            rid = ln[1:]
            seq = lines[i + 1].rstrip(b"\r")
            i += 4
            recs.append((rid, seq))
        else:
            raise ValueError("bad fastx line: %r" % ln)
    return recs


def read_sam(path):
    """-> (header_lines: list[bytes], records: list[list[bytes]] (tab-split fields))"""
    hdr, recs = [], []
    with _open(path) as f:
        for ln in f.read().split(b"\n"):
            if not ln:
                continue
            if ln[:1] == b"@":
                hdr.append(ln)
            else:
                recs.append(ln.split(b"\t"))
    return hdr, recs

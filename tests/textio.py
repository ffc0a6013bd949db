"""Tiny FASTA/FASTQ/SAM readers used by the tests to turn golden files into record lists.
Test infrastructure; follows needletail's observable behaviour as listed in SURVEY.md §5
(id = header line minus the marker, seq = lines joined without newlines)."""
import gzip
import struct

import numpy as np


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


def bam_like(n_rec, seed=5, const_qual=False):
    """records shaped like `tag`'s BAM output: fixed fields, a counting name, nibble-packed random bases, qualities, a tag"""
    rng = np.random.default_rng(seed)
    L = 150
    rec = np.zeros((n_rec, 36 + 12 + 4 + L // 2 + L + 14), dtype=np.uint8)
    rec[:, 0:4] = np.frombuffer(struct.pack("<I", rec.shape[1] - 4), dtype=np.uint8)
    pos = (np.arange(n_rec, dtype=np.uint32) * 37) % 2000000
    rec[:, 8:12] = pos.view(np.uint8).reshape(n_rec, 4)
    rec[:, 12] = 12
    rec[:, 13] = 60
    rec[:, 16] = 1
    rec[:, 20:24] = np.frombuffer(struct.pack("<I", L), dtype=np.uint8)
    rec[:, 24:28] = 255
    names = np.array([b"r%010d\0" % i for i in range(n_rec)], dtype="S12")
    rec[:, 36:48] = names.view(np.uint8).reshape(n_rec, 12)
    rec[:, 48:52] = np.frombuffer(struct.pack("<I", L << 4), dtype=np.uint8)
    nib = np.array([1, 2, 4, 8], dtype=np.uint8)[rng.integers(0, 4, size=(n_rec, L))]
    rec[:, 52:52 + L // 2] = nib[:, 0::2] << 4 | nib[:, 1::2]
    q0 = 52 + L // 2
    rec[:, q0:q0 + L] = 40 if const_qual else np.array([2, 12, 23, 37], dtype=np.uint8)[rng.choice(4, size=(n_rec, L), p=[0.02, 0.05, 0.13, 0.8])]
    rec[:, q0 + L:] = np.frombuffer(b"NMC\0ASC\x96XSZabc\0", dtype=np.uint8)[:14]
    return rec.tobytes()

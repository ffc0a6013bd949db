"""FASTQ text -> records on the device (mk_extract_fastq_text, ingest.hip) against the host path:
the same loop results (keep, log rows, counters) as mk_extract_single on the sequences a host parser extracts from the
same text, the record table (where every record starts), and the refusal (status 1) of everything that is not plain
4-line FASTQ -- which the caller then parses with its own reader, the byte-identical checker of this path.
needletail's FASTQ reader (src/cmd_extract.rs:281-282): 4-line records, '@' may start a quality line, LF or CRLF."""
import random

import numpy as np
import pytest

import oracle_binding as ob

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mk():
    from merkurio_amd import native
    native.load()
    if native.device_count() < 1:
        pytest.fail("no HIP device visible: the gpu-marked tests need an MI355X")
    return native


def _rand(rnd, n, alpha=b"ACGT"):
    return bytes(rnd.choice(alpha) for _ in range(n))


def _fastq(recs, eol=b"\n", plus_ids=False, final_eol=True):
    out = []
    for i, (rid, seq, qual) in enumerate(recs):
        out.append(b"@" + rid + eol + seq + eol + (b"+" + rid if plus_ids and i % 3 == 0 else b"+") + eol + qual + eol)
    text = b"".join(out)
    return text if final_eol else text[:-len(eol)]


def _case(seed, n, lens, patterns, qual_alpha=b"@+IJ#5ACGT>"):
    rnd = random.Random(seed)
    recs = []
    for i in range(n):
        L = rnd.choice(lens)
        s = bytearray(_rand(rnd, L))
        if patterns and L >= 31 and rnd.random() < 0.2:
            p = rnd.choice(patterns)
            k = rnd.randrange(0, L - len(p) + 1)
            s[k:k + len(p)] = p
        # quality strings that start with '@' or '+' and contain sequence letters: lines are counted, not guessed
        q = _rand(rnd, L, qual_alpha)
        if L and i % 5 == 0:
            q = b"@" + q[1:]
        if L and i % 7 == 0:
            q = b"+" + q[1:]
        recs.append((b"r%d some description" % i, bytes(s), q))
    return recs


@pytest.mark.parametrize("lens,eol,final_eol,plus_ids", [
    ([150], b"\n", True, False),             # reads of one length: no offsets at all on the device
    ([150], b"\r\n", True, True),            # CRLF, '+id' third lines
    ([36, 75, 150, 151, 250], b"\n", False, True),  # trimmed reads, no final newline
    ([0, 1, 31, 150], b"\n", True, False),   # empty and tiny reads
    ([31], b"\r\n", False, False),
])
def test_fastq_text_equals_host_path(mk, lens, eol, final_eol, plus_ids):
    rnd = random.Random(len(lens) * 100 + len(eol))
    patterns = mk.parse_pattern_list(kmer_seq=[_rand(rnd, 31) for _ in range(300)], reverse_complement=True)
    recs = _case(7 + len(lens), 20_000, lens, patterns)
    text = _fastq(recs, eol, plus_ids, final_eol)
    seqs = [s for _, s, _ in recs]
    m = mk.Matcher(patterns)
    om = ob.Matcher(patterns, True, 0, False)
    for logging in (True, False):
        for invert in (False, True):
            status, rec_start, keep, rows, c = m.extract_fastq_text(text, logging=logging, invert=invert)
            assert status == 0
            k_o, r_o, c_o = ob.extract_single(om, seqs, logging=logging, invert=invert)
            assert keep == k_o and rows == r_o and c == c_o
            assert (keep, rows, c) == m.extract_single(seqs, logging=logging, invert=invert)
            # the record table: every record starts at its '@', the last entry is the end of the text
            assert len(rec_start) == len(recs) + 1 and rec_start[-1] == len(text)
            pos = 0
            for i, (rid, s, q) in enumerate(recs[:2000]):
                assert rec_start[i] == pos and text[pos:pos + 1] == b"@"
                pos = text.index(b"@" + recs[i + 1][0], pos + 1) if i + 1 < len(recs) else len(text)
    assert sum(keep) > 1000


def test_text_that_is_not_plain_fastq_is_refused(mk):
    """status 1 and no output for: FASTA, a blank line between records, a wrapped (multi-line) record, a quality string
    of another length, a missing '+' line, a truncated last record; the empty text is zero records"""
    m = mk.Matcher([b"ACGTACGTACGTACGTACGTACGTACGTACG"])
    rec = lambda i, s=b"ACGT" * 10, q=None: b"@r%d\n" % i + s + b"\n+\n" + (q if q is not None else b"I" * len(s)) + b"\n"
    good = b"".join(rec(i) for i in range(8))
    assert m.extract_fastq_text(good)[0] == 0 and len(m.extract_fastq_text(good)[2]) == 8
    assert m.extract_fastq_text(b"")[:3] == (0, [0], [])
    bad = {
        "fasta": b">a\nACGT\n>b\nACGT\n>c\nAC\n>d\nAA\n",
        "blank line": rec(0) + b"\n" + rec(1) + rec(2) + rec(3)[:-1],
        "wrapped": b"@w\nACGT\nACGT\n+\nIIIIIIII\n" + rec(1) + rec(2) + rec(3)[:-len(b"IIII\n") - 36] ,
        "quality length": rec(0) + rec(1, q=b"I" * 39) + rec(2),
        "no plus": rec(0) + b"@x\nACGT\n-\nIIII\n",
        "truncated": good + b"@t\nACGT\n+\n",
        "not a header": rec(0) + b"r1\nACGT\n+\nIIII\n",
    }
    for name, text in bad.items():
        status, rec_start, keep, rows, c = m.extract_fastq_text(text)
        assert status == 1 and keep == [] and rows == [], name


def test_fastq_text_large_window(mk):
    """a 300 MB window (2 M reads of 100 bp, a third of them with a k-mer): block scans across thousands of 16 KiB
    blocks and record tiles, against the host path"""
    rng = np.random.default_rng(3)
    n, L = 2_000_000, 100
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    pats = [acgt[rng.integers(0, 4, 31)].tobytes() for _ in range(1000)]
    patterns = mk.parse_pattern_list(kmer_seq=pats)
    bases = acgt[rng.integers(0, 4, size=(n, L))]
    for i in range(0, n, 3):
        bases[i, 11:42] = np.frombuffer(patterns[i % len(patterns)], dtype=np.uint8)
    H = 10
    rec = np.empty((n, H + L + 3 + L + 1), dtype=np.uint8)
    rec[:, :H] = np.array([f"@{i:08d}\n" for i in range(n)], dtype="S10").view(np.uint8).reshape(n, H)
    rec[:, H:H + L] = bases
    rec[:, H + L:H + L + 3] = np.frombuffer(b"\n+\n", dtype=np.uint8)
    rec[:, H + L + 3:H + 2 * L + 3] = ord("@")  # every quality line starts with '@'
    rec[:, -1] = ord("\n")
    text = rec.tobytes()
    m = mk.Matcher(patterns)
    status, rec_start, keep, rows, c = m.extract_fastq_text(text, logging=True)
    assert status == 0 and len(keep) == n
    assert np.array_equal(np.asarray(rec_start), np.arange(n + 1, dtype=np.uint64) * rec.shape[1])
    seqs = [bases[i].tobytes() for i in range(n)]
    k2, r2, c2 = m.extract_single(seqs, logging=True)
    assert keep == k2 and rows == r2 and c == c2 and sum(keep) >= n // 3

"""mk_extract_fastq_bgzf (include/merkurio_hip.h v5): a window of a bgzip'ed FASTQ inflated on the device straight into the text
buffer the ingest kernels read, against mk_extract_fastq_text on the same text (itself checked against the host path and the
oracle in test_gpu_ingest.py) -- window by window, with the unfinished record of a window carried as the next one's head, as a
caller does.  Replaces needletail's gzip reader + record loop for bgzip'ed input (src/cmd_extract.rs:281-282,321-328)."""
import random

import numpy as np
import pytest

from merkurio_amd import native as mk
from test_gpu_codec import zlib_bgzf

pytestmark = pytest.mark.gpu


def fastq(rng, n, kmers, crlf=False, final_newline=True, lens=(150,)):
    nl = b"\r\n" if crlf else b"\n"
    recs = []
    for i in range(n):
        L = rng.choice(lens)
        s = bytearray(rng.choice(b"ACGT") for _ in range(L))
        if i % 5 == 0 and L >= 40:
            k = rng.choice(kmers)
            o = rng.randrange(L - len(k) + 1)
            s[o:o + len(k)] = k
        q = bytes(rng.choice(b"@+FFFF:#") for _ in range(L))  # '@' / '+' open some quality lines
        recs.append(b"@r%d extra" % i + nl + bytes(s) + nl + b"+" + nl + q + nl)
    text = b"".join(recs)
    return text if final_newline else text[:-len(nl)]


@pytest.fixture(scope="module")
def setup():
    rng = random.Random(77)
    kmers = [bytes(rng.choice(b"ACGT") for _ in range(31)) for _ in range(50)]
    m = mk.Matcher(kmers, device=0)
    c = mk.Codec(0)
    yield rng, kmers, m, c
    c.close()


def through_windows(m, c, text, members_per_window, block=65280, whole_text=True, **kw):
    """the whole file in windows of `members_per_window` members -> (keep per record, rows with file-wide record numbers, counters summed
    [, the kept records' text as it came back, whole_text=False: only that and each window's unfinished tail come back])"""
    blob = zlib_bgzf(text, block, level=6)
    mem, used, total = mk.bgzf_members(blob)
    assert used == len(blob) and total == len(text)
    head, keep, rows, rec0, at = b"", [], [], 0, 0
    sums = None
    got_text, got_kept = b"", b""
    while at < len(mem) or head:
        group = mem[at:at + members_per_window]
        at += len(group)
        last = at >= len(mem)
        status, wtext, n_used, rec_start, k, r, cnt = m.extract_fastq_bgzf(c, head, blob, group, last, whole_text=whole_text, **kw)
        assert status == 0, status
        if whole_text:
            assert wtext[:len(head)] == head
            got_text += wtext[len(head):]
        else:
            tail, kept_text = wtext
            assert len(kept_text) == sum(rec_start[i + 1] - rec_start[i] for i in range(len(k)) if k[i])
            got_kept += kept_text
            wtext = b"\0" * n_used + tail  # (what follows only needs the tail behind n_used)
        assert rec_start[-1] == n_used if rec_start else n_used == 0
        keep += k
        rows += [(f, rec0 + rec, p, pos) for f, rec, p, pos in r]
        rec0 += len(k)
        if sums is None:
            sums = cnt
        else:
            for key in ("records", "bases", "extracted"):
                sums[key] += cnt[key]
            sums["hits"] = tuple(a + b for a, b in zip(sums["hits"], cnt["hits"]))
            sums["records_hit"] = tuple(a + b for a, b in zip(sums["records_hit"], cnt["records_hit"]))
            sums["pattern_hit_counts"] = [a + b for a, b in zip(sums["pattern_hit_counts"], cnt["pattern_hit_counts"])]
        head = wtext[n_used:]
        if last:
            assert head == b""
            break
    if not whole_text:
        return keep, rows, sums, got_kept
    assert got_text == text
    return keep, rows, sums


@pytest.mark.parametrize("flavour", ["plain", "crlf", "ragged", "no final newline"])
def test_windows_of_members_equal_the_text_entry(setup, flavour):
    rng, kmers, m, c = setup
    text = fastq(rng, 6000, kmers, crlf=flavour == "crlf", final_newline=flavour != "no final newline",
                 lens=(150,) if flavour != "ragged" else (36, 75, 100, 151))
    status, rec_start, keep0, rows0, cnt0 = m.extract_fastq_text(text, logging=True)
    assert status == 0 and sum(keep0) >= 800
    for per_window, block in ((1000, 65280), (3, 65280), (1, 20000), (7, 777)):
        keep, rows, cnt = through_windows(m, c, text, per_window, block, logging=True)
        assert keep == keep0 and rows == rows0 and cnt == cnt0, (flavour, per_window, block)
    keep, rows, cnt = through_windows(m, c, text, 5, 30000, logging=False, invert=True)
    assert keep == [not k for k in keep0] and rows == []
    # only the kept records and each window's unfinished tail come back (gathered on the device): the same results, and the
    # kept text is the kept records of the file, back to back
    want_kept = b"".join(text[rec_start[i]:rec_start[i + 1]] for i in range(len(keep0)) if keep0[i])
    for per_window, block in ((1000, 65280), (2, 20000), (9, 777)):
        keep, rows, cnt, kept_text = through_windows(m, c, text, per_window, block, whole_text=False, logging=True)
        assert keep == keep0 and rows == rows0 and cnt == cnt0 and kept_text == want_kept, (flavour, per_window, block)
    keep, rows, cnt, kept_text = through_windows(m, c, text, 4, 30000, whole_text=False, logging=False, invert=True)
    assert keep == [not k for k in keep0] and kept_text == b"".join(text[rec_start[i]:rec_start[i + 1]] for i in range(len(keep0)) if not keep0[i])


def test_refusals_and_damage(setup):
    rng, kmers, m, c = setup
    good = fastq(rng, 300, kmers)
    for bad in (good.replace(b"\n+\n", b"\n\n+\n", 1),          # a blank line
                b">a\nACGT\n>b\nACGT\n" * 50,                    # FASTA
                good[:-200]):                                       # an unfinished record at the end of the input
        blob = zlib_bgzf(bad, 65280, level=1)
        mem, _, _ = mk.bgzf_members(blob)
        status = m.extract_fastq_bgzf(c, b"", blob, mem, True)[0]
        assert status == 1
    # a window that ends inside a record is no refusal as long as text follows: the tail comes back as the next head
    blob = zlib_bgzf(good, 10000, level=6)
    mem, _, _ = mk.bgzf_members(blob)
    status, wtext, n_used, rec_start, keep, rows, cnt = m.extract_fastq_bgzf(c, b"", blob, mem[:2], False)
    assert status == 0 and 0 < n_used < len(wtext) == 20000 and wtext == good[:20000] and good[n_used:n_used + 1] == b"@" and good[n_used - 1:n_used] == b"\n"
    # damage
    dmg = bytearray(blob)
    dmg[int(mem[1]["data_off"]) + 30] ^= 0x55
    with pytest.raises(mk.MerkurioError) as e:
        m.extract_fastq_bgzf(c, b"", bytes(dmg), mem, True)
    assert e.value.code == mk.MK_E_CORRUPT and "member 1" in str(e.value)
    # an empty input, and only an end-of-file member
    assert m.extract_fastq_bgzf(c, b"", b"", mem[:0], True)[:3] == (0, b"", 0)
    eof = mk.bgzf_eof()
    em, _, _ = mk.bgzf_members(eof)
    assert m.extract_fastq_bgzf(c, b"", eof, em, True)[:3] == (0, b"", 0)


def test_a_large_window(setup):
    rng, kmers, m, c = setup
    unit = fastq(rng, 3000, kmers)
    text = unit * 150  # 150 MB... 3000 x ~320 B x 150 = 144 MB, 2 200 members
    blob = c.deflate(text)  # (the device writes the members: zlib would take a minute here)
    mem, _, _ = mk.bgzf_members(blob)
    status, wtext, n_used, rec_start, keep, rows, cnt = m.extract_fastq_bgzf(c, b"", blob, mem, True, logging=False)
    assert status == 0 and n_used == len(text) and wtext == text and len(keep) == 3000 * 150
    s0, r0, keep0, _, cnt0 = m.extract_fastq_text(unit, logging=False)
    assert keep == keep0 * 150 and cnt["extracted"] == cnt0["extracted"] * 150


def test_fuzz_windows(setup):
    """seeded: read flavours x member sizes x members per window x what comes back (whole text / tail + kept records) x logging / -v --
    always the results of mk_extract_fastq_text on the same text"""
    import os
    rng, kmers, m, c = setup
    for seed in range(int(os.environ.get("MERKURIO_FUZZ_BGZF", "24"))):
        r = random.Random(500 + seed)
        text = fastq(r, r.randrange(1, 1500), kmers, crlf=r.random() < 0.25, final_newline=r.random() < 0.7,
                     lens=r.choice(((150,), (36, 75, 100, 151), (31, 32, 40), (250,))))
        logging = r.random() < 0.6
        invert = r.random() < 0.3
        status, rec_start, keep0, rows0, cnt0 = m.extract_fastq_text(text, logging=logging, invert=invert)
        assert status == 0
        per_window, block = r.choice((1, 2, 5, 50, 10000)), r.choice((65280, r.randrange(600, 65281)))
        whole = invert and logging or r.random() < 0.4
        res = through_windows(m, c, text, per_window, block, whole_text=whole, logging=logging, invert=invert)
        assert res[0] == keep0 and res[1] == rows0 and res[2] == cnt0, (seed, per_window, block, whole, logging, invert)
        if not whole:
            assert res[3] == b"".join(text[rec_start[i]:rec_start[i + 1]] for i in range(len(keep0)) if keep0[i]), seed

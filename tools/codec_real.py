#!/usr/bin/env python3
"""The device inflater on members a real writer made (r05): zlib LEVEL-6 BGZF members (what htslib / bgzip write) of
  (a) BAM records, 150 bp, 4-bin qualities (tests/textio.py: bam_like) and
  (b) FASTQ, 150 bp, Illumina-style binned qualities in runs (tools/e2e_pairs.py's model),
at 64 MB ... 2 GB of text per call, next to the same text in members the device's own deflate wrote (long matches, few tokens:
the friendly input the r04 figures were measured on).  kernel ms = hipEvents around mk_bgzf_inflate_kernel + the CRC check.
usage: tools/codec_real.py [largest size in MB, default 2048]"""
import os, sys, time, zlib
from concurrent.futures import ThreadPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from merkurio_amd import native as mk
from textio import bam_like

top = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
rng = np.random.default_rng(11)


def fastq_text(n, L=150):
    bins = np.frombuffer(b"FFF:,#", dtype=np.uint8)
    q = np.empty((n, L), dtype=np.uint8)
    cur = rng.integers(0, 2, size=n).astype(np.uint8)
    for j in range(L):
        change = rng.random(n) < 0.08
        nxt = rng.integers(0, 3 + (3 * j) // L, size=n).astype(np.uint8)
        cur = np.where(change, nxt, cur)
        q[:, j] = bins[cur]
    H = 13
    rec = np.empty((n, H + L + 3 + L + 1), dtype=np.uint8)
    rec[:, :H] = np.array([f"@r{i:010d}\n" for i in range(n)], dtype="S13").view(np.uint8).reshape(n, H)
    rec[:, H:H + L] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n, L))]
    rec[:, H + L:H + L + 3] = np.frombuffer(b"\n+\n", dtype=np.uint8)
    rec[:, H + L + 3:H + 2 * L + 3] = q
    rec[:, -1] = ord("\n")
    return rec.tobytes()


def bgzf6(raw):
    def member(b):
        chunk = raw[b:b + 0xff00]
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        c = co.compress(chunk) + co.flush()
        return (bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0]) + (len(c) + 25).to_bytes(2, "little") + c +
                zlib.crc32(chunk).to_bytes(4, "little") + len(chunk).to_bytes(4, "little"))
    with ThreadPoolExecutor(16) as ex:
        return b"".join(ex.map(member, range(0, len(raw), 0xff00)))


codec = mk.Codec()
for label, make in (("BAM records, 4-bin qualities", lambda mb: bam_like(mb * (1 << 20) // 268 + 1, seed=3)[:mb << 20]),
                    ("FASTQ, binned qualities in runs", lambda mb: fastq_text(mb * (1 << 20) // 317 + 1)[:mb << 20])):
    t0 = time.time()
    unit_mb = min(top, 512)
    unit = make(unit_mb)
    z_unit = bgzf6(unit)
    d_unit = codec.deflate(unit)
    print(f"{label}: {len(unit) / 1e6:.0f} MB unit, zlib level 6 ratio {len(unit) / len(z_unit):.2f}, device deflate ratio {len(unit) / len(d_unit):.2f} "
          f"({time.time() - t0:.0f} s to make)", flush=True)
    mem_z, _, _ = mk.bgzf_members(z_unit)
    for mb in (64, 256, 1024, 2048):  # (2048: the lane-per-member kernel and the smallest rings only)
        if mb > top:
            continue
        for kind, blob_unit in (("zlib level 6 members", z_unit), ("device-written members", d_unit)):
            if mb <= unit_mb:
                # whole members only: the first members that hold mb MB of text
                tab, _, _ = mk.bgzf_members(blob_unit)
                cum = np.cumsum(tab["isize"].astype(np.int64))
                k = int(np.searchsorted(cum, mb << 20)) + 1
                k = min(k, len(tab))
                end = int(tab["data_off"][k - 1]) + int(tab["data_len"][k - 1]) + 8
                blob, want = blob_unit[:end], unit[:int(cum[k - 1])]
            else:
                reps = mb // unit_mb
                blob, want = blob_unit * reps, None
            for which, kname in ((0, "chosen by size"), (1, "lane per member"), (2, "wave, 32 KiB ring"), (4, "wave, 16 KiB ring"), (3, "wave,  8 KiB ring"), (5, "wave,  4 KiB ring"), (6, "wave,  2 KiB ring")):
                if which in (2, 4) and mb > 256 or which == 3 and mb > 1024:
                    continue
                codec.set_inflate_kernel(which)
                best = None
                for r in range(3):
                    t1 = time.time()
                    text = codec.inflate(blob)
                    dt = codec.last_call_s  # (the C call alone)
                    up, dev, down = codec.times()
                    if best is None or dev < best[1]:
                        best = (dt, dev, up, down)
                if want is not None:
                    assert text == want
                else:
                    assert len(text) == len(unit) * reps and text[:1 << 20] == unit[:1 << 20] and text[-(1 << 20):] == unit[-(1 << 20):]
                n_text = len(text)
                del text
                print(f"  {mb:5d} MB, {kind:24s}, {kname:16s}: kernels {best[1]:7.2f} ms = {n_text / best[1] / 1e6:6.1f} GB/s of text; call {best[0] * 1e3:6.0f} ms "
                      f"(upload {best[2]:.0f}, download {best[3]:.0f})", flush=True)
            codec.set_inflate_kernel(0)

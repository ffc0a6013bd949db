#!/usr/bin/env python3
"""End-to-end wall time of `merkurio extract` on BASELINE's own input shapes (r05): paired FASTQ (config 3's shape), plain and
bgzip'ed with zlib LEVEL-6 members and Illumina-style binned qualities (not constant: the r04 figures were on level-1 members of
constant qualities), and a genome-like FASTA -- through the text windows the device indexes (extract_windows.cpp, the default)
and through the host reader (--host-ingest), and dealt to two handles (--gpus 2; on a one-GPU box both share the device).
usage: tools/e2e_pairs.py [pairs, default 10 000 000] [patterns, default 10 000] [fasta megabases, default 1000]"""
import gc
import os
import subprocess
import sys
import time
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
npat = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
fa_mb = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
L = 150
tmp = os.environ.get("TMPDIR", "/tmp")
rng = np.random.default_rng(3)
binp = os.path.join(ROOT, "merkurio_amd", "lib", "merkurio")
km = os.path.join(tmp, "e2ep_kmers.txt")
pats = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(npat, 31))]
open(km, "wb").write(b"\n".join(p.tobytes() for p in pats) + b"\n")
EOF_MARK = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def binned_qualities(n, L):
    """NovaSeq-style 4-bin qualities with runs: a position keeps its predecessor's bin with p = 0.92, quality falls towards the 3' end"""
    bins = np.frombuffer(b"FFF:,#", dtype=np.uint8)
    q = np.empty((n, L), dtype=np.uint8)
    cur = rng.integers(0, 2, size=n).astype(np.uint8)
    for j in range(L):
        change = rng.random(n) < 0.08
        nxt = rng.integers(0, 3 + (3 * j) // L, size=n).astype(np.uint8)  # the lower bins open up along the read
        cur = np.where(change, nxt, cur)
        q[:, j] = bins[cur]
    return q


def fastq(n, tag, seed_shift):
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n, L))]
    for i in range(seed_shift, n, 100):  # 1 % of the reads of each file carry a k-mer
        bases[i, 7:38] = pats[i % npat]
    H = 15
    rec = np.empty((n, H + L + 3 + L + 1), dtype=np.uint8)
    hdr = np.array([f"@r{i:010d}/{tag}\n" for i in range(n)], dtype=f"S{H}")
    rec[:, :H] = hdr.view(np.uint8).reshape(n, H)
    rec[:, H:H + L] = bases
    rec[:, H + L:H + L + 3] = np.frombuffer(b"\n+\n", dtype=np.uint8)
    rec[:, H + L + 3:H + 2 * L + 3] = binned_qualities(n, L)
    rec[:, -1] = ord("\n")
    return rec


def bgzf(raw, level=6):
    def member(b):
        chunk = raw[b:b + 0xff00]
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        c = co.compress(chunk) + co.flush()
        return (bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0]) + (len(c) + 25).to_bytes(2, "little") + c +
                zlib.crc32(chunk).to_bytes(4, "little") + len(chunk).to_bytes(4, "little"))
    with ThreadPoolExecutor(16) as ex:
        return b"".join(ex.map(member, range(0, len(raw), 0xff00))) + EOF_MARK


t0 = time.time()
paths = {}
for tag in (1, 2):
    rec = fastq(n, tag, 50 * (tag - 1))
    p = os.path.join(tmp, f"e2ep_{tag}.fastq")
    rec.tofile(p)
    raw = rec.tobytes()
    del rec
    gc.collect()
    open(p + ".gz", "wb").write(bgzf(raw))
    paths[tag] = p
    print(f"mate {tag}: {n} reads, {len(raw) / 1e6:.0f} MB FASTQ, {os.path.getsize(p + '.gz') / 1e6:.0f} MB as BGZF (zlib level 6, binned qualities: "
          f"ratio {len(raw) / os.path.getsize(p + '.gz'):.2f}), {time.time() - t0:.0f} s", flush=True)
    del raw
    gc.collect()


def run(label, args, outs, bases):
    t0 = time.time()
    subprocess.run([binp, "extract", *args, "-f", km, "-o", os.path.join(tmp, "e2ep_out")], check=True, env=dict(os.environ, MERKURIO_TIMING="0"))
    dt = time.time() - t0
    sizes = [os.path.getsize(os.path.join(tmp, o)) for o in outs]
    print(f"{label}: {dt:.2f} s wall -> {bases / dt / 1e9:.2f} Gbases/s end to end; output {'+'.join(str(s) for s in sizes)} bytes", flush=True)
    return sizes


pair_outs = ["e2ep_out_1.fastq", "e2ep_out_2.fastq"]
for rep in range(2):  # (the first process on a fresh box also pays the GPU's start-up)
    a = run(f"paired plain 2 x {n}, device windows", ["-i", paths[1], "-2", paths[2]], pair_outs, 2 * n * L)
    b = run(f"paired plain 2 x {n}, --host-ingest", ["-i", paths[1], "-2", paths[2], "--host-ingest"], pair_outs, 2 * n * L)
    assert a == b
c = run(f"paired plain 2 x {n}, device windows, --gpus 2", ["-i", paths[1], "-2", paths[2], "--gpus", "2"], pair_outs, 2 * n * L)
assert c == a
for rep in range(2):
    z = run(f"paired BGZF (level 6) 2 x {n}, device windows (members inflated on the device)", ["-i", paths[1] + ".gz", "-2", paths[2] + ".gz"], pair_outs, 2 * n * L)
    assert z == a
zh = run(f"paired BGZF 2 x {n}, device windows, --host-codec (zlib on the host threads)", ["-i", paths[1] + ".gz", "-2", paths[2] + ".gz", "--host-codec"], pair_outs, 2 * n * L)
zz = run(f"paired BGZF 2 x {n}, --host-ingest --host-codec (the r03 path)", ["-i", paths[1] + ".gz", "-2", paths[2] + ".gz", "--host-ingest", "--host-codec"], pair_outs, 2 * n * L)
assert zh == a and zz == a
s1 = run(f"single BGZF (level 6) {n}, device windows", ["-i", paths[1] + ".gz"], ["e2ep_out.fastq"], n * L)
s2 = run(f"single BGZF (level 6) {n}, --host-ingest --host-codec", ["-i", paths[1] + ".gz", "--host-ingest", "--host-codec"], ["e2ep_out.fastq"], n * L)
assert s1 == s2
for p in paths.values():
    os.remove(p), os.remove(p + ".gz")

# ---- a genome-like FASTA: 24 records of fa_mb / 24 Mbp, wrapped at 60 columns; 10 k 31-mers sampled from it (every record kept)
t0 = time.time()
fa = os.path.join(tmp, "e2ep_genome.fa")
per = fa_mb * 1_000_000 // 24 // 60 * 60
with open(fa, "wb") as f:
    picks = []
    for c in range(24):
        s = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=per)]
        for _ in range(npat // 24 + 1):
            o = int(rng.integers(0, per - 31))
            picks.append(s[o:o + 31].tobytes())
        lines = np.empty((per // 60, 61), dtype=np.uint8)
        lines[:, :60] = s.reshape(-1, 60)
        lines[:, 60] = ord("\n")
        f.write(b">chr%d synthetic\n" % (c + 1))
        f.write(lines.tobytes())
open(km, "wb").write(b"\n".join(picks[:npat]) + b"\n")
print(f"FASTA: 24 records x {per / 1e6:.1f} Mbp, {os.path.getsize(fa) / 1e6:.0f} MB, {time.time() - t0:.0f} s", flush=True)
for rep in range(2):
    f1 = run("genome FASTA, device windows", ["-i", fa], ["e2ep_out.fa"], 24 * per)
    f2 = run("genome FASTA, --host-ingest", ["-i", fa, "--host-ingest"], ["e2ep_out.fa"], 24 * per)
    assert f1 == f2
os.remove(fa)

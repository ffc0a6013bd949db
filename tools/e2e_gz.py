#!/usr/bin/env python3
"""End-to-end `merkurio extract` on plain-gzip FASTQ (one member, what `gzip` / `pigz` write; the tool's usual input): the member is
inflated in parallel pieces on the device (default) against zlib on the host (--host-codec) and against the r03 host path
(--host-ingest --host-codec); single and paired.  usage: tools/e2e_gz.py [reads, default 4 000 000] [gzip level, default 6]"""
import os, subprocess, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from bench import _fastq_binned
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
level = int(sys.argv[2]) if len(sys.argv) > 2 else 6
tmp = os.environ.get("TMPDIR", "/tmp")
binp = os.path.join(ROOT, "merkurio_amd", "lib", "merkurio")
rng = np.random.default_rng(9)
pats = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(10000, 31))]
km = os.path.join(tmp, "e2egz_kmers.txt")
paths = []
for tag in (1, 2):
    raw = bytearray(_fastq_binned(n, seed=20 + tag))
    rec = 13 + 150 + 3 + 150 + 1
    for i in range(0, n, 100):  # 1 % of the reads carry a k-mer
        raw[i * rec + 13 + 7:i * rec + 13 + 38] = pats[i % 10000].tobytes()
    raw = bytes(raw)
    p = os.path.join(tmp, f"e2egz_{tag}.fastq.gz")
    t0 = time.time()
    co = zlib.compressobj(level, zlib.DEFLATED, 31)
    open(p, "wb").write(co.compress(raw) + co.flush())
    print(f"mate {tag}: {n} reads, {len(raw) / 1e6:.0f} MB of FASTQ -> {os.path.getsize(p) / 1e6:.0f} MB (gzip -{level}, one member, {time.time() - t0:.0f} s)", flush=True)
    paths.append(p)
    del raw
open(km, "wb").write(b"\n".join(p.tobytes() for p in pats) + b"\n")


def run(label, args, outs, bases, timing=False):
    t0 = time.time()
    env = dict(os.environ)
    if timing:
        env["MERKURIO_TIMING"] = "1"
    r = subprocess.run([binp, "extract", *args, "-f", km, "-o", os.path.join(tmp, "e2egz_out")], env=env, capture_output=True, text=True)
    dt = time.time() - t0
    assert r.returncode == 0, r.stderr[-2000:]
    sizes = [os.path.getsize(os.path.join(tmp, o)) for o in outs]
    gz = [ln.strip() for ln in r.stderr.splitlines() if "gzip input" in ln or (os.environ.get("E2E_ALL_TIMING") and "[timing]" in ln)]
    print(f"{label}: {dt:.2f} s wall -> {bases / dt / 1e9:.2f} Gbases/s end to end; output {'+'.join(map(str, sizes))} bytes" + ("".join("\n      " + g for g in gz)), flush=True)
    return sizes


one = ["e2egz_out.fastq"]
for rep in range(2):
    a = run(f"single .fastq.gz, {n} reads: inflated in parallel pieces on the device", ["-i", paths[0]], one, n * 150, timing=rep == 1)
    b = run(f"single .fastq.gz, {n} reads: --host-codec (zlib on one host thread feeds the device windows)", ["-i", paths[0], "--host-codec"], one, n * 150)
    assert a == b
c = run(f"single .fastq.gz, {n} reads: --host-ingest --host-codec (the r03 path)", ["-i", paths[0], "--host-ingest", "--host-codec"], one, n * 150)
assert c == a
two = ["e2egz_out_1.fastq", "e2egz_out_2.fastq"]
for rep in range(2):
    p1 = run(f"paired .fastq.gz, 2 x {n} reads: both members on the device", ["-i", paths[0], "-2", paths[1]], two, 2 * n * 150, timing=rep == 1)
    p2 = run(f"paired .fastq.gz, 2 x {n} reads: --host-codec", ["-i", paths[0], "-2", paths[1], "--host-codec"], two, 2 * n * 150)
    assert p1 == p2
for p in paths:
    os.remove(p)

#!/usr/bin/env python3
"""End-to-end wall time of the C++ CLI on a synthetic FASTQ (ingest + PCIe + scan + write): the default path (a single
FASTQ is uploaded as text and indexed on the GPU), the host-parser path (--host-ingest) and, with MERKURIO_E2E_GZ=1, the
same reads gzip'd (one member, zlib streaming) and BGZF'd (64 KiB members, inflated on all host threads).
usage: tools/e2e_cli.py [n_reads] [n_patterns] [one read in N carries a k-mer, default 100]"""
import os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
npat = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
every = int(sys.argv[3]) if len(sys.argv) > 3 else 100
L = 150
rng = np.random.default_rng(1)
tmp = os.environ.get("TMPDIR", "/tmp")
fq, km = os.path.join(tmp, "e2e.fastq"), os.path.join(tmp, "e2e_kmers.txt")
t0 = time.time()
bases = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n, L))]
pats = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(npat, 31))]
if every == 1:
    bases[:, 7:38] = pats[np.arange(n) % npat]
else:
    for i in range(0, n, every):  # 1 % of the reads carry a k-mer by default
        bases[i, 7:38] = pats[i % npat]
H = 13
rec = np.empty((n, H + L + 3 + L + 1), dtype=np.uint8)
hdr = np.array([f"@r{i:010d}\n" for i in range(n)], dtype="S13")
rec[:, :H] = hdr.view(np.uint8).reshape(n, H)
rec[:, H:H + L] = bases
rec[:, H + L:H + L + 3] = np.frombuffer(b"\n+\n", dtype=np.uint8)
rec[:, H + L + 3:H + 2 * L + 3] = ord("I")
rec[:, -1] = ord("\n")
rec.tofile(fq)
open(km, "wb").write(b"\n".join(p.tobytes() for p in pats) + b"\n")
print(f"generated {n} reads ({os.path.getsize(fq) / 1e6:.0f} MB FASTQ) in {time.time() - t0:.1f} s", flush=True)
GZ = bool(os.environ.get("MERKURIO_E2E_GZ"))
m_small = min(n, int(os.environ.get("MERKURIO_E2E_GZ_READS", "4000000")))
raw_small = rec[:m_small].tobytes() if GZ else None
# the timed runs are children of this process: drop its 12 GB of arrays first (a fork of a large parent is not free)
del rec, bases, hdr
import gc
gc.collect()
binp = os.path.join(ROOT, "merkurio_amd", "lib", "merkurio")
LOGS = ["-l", os.path.join(tmp, "e2e.log"), "-j", os.path.join(tmp, "e2e.json")]


def run(label, path, extra, n_reads):
    t0 = time.time()
    subprocess.run([binp, "extract", "-i", path, "-f", km, "-o", os.path.join(tmp, "e2e_out"), *extra], check=True,
                   env=dict(os.environ, MERKURIO_TIMING="1"))
    dt = time.time() - t0
    kept = os.path.getsize(os.path.join(tmp, "e2e_out.fastq")) // (13 + 2 * L + 4)
    logs = sum(os.path.getsize(os.path.join(tmp, f)) for f in ("e2e.log", "e2e.json") if "-l" in extra and os.path.exists(os.path.join(tmp, f)))
    print(f"{label}: {dt:.2f} s wall -> {n_reads * L / dt / 1e9:.3f} Gbases/s end to end, {kept} reads extracted" + (f", {logs / 1e6:.0f} MB of logs" if logs else ""), flush=True)
    return kept


for rep in range(2):  # (the first process on a fresh box also pays the GPU's start-up)
    k0 = run("extract (no log), device ingest", fq, [], n)
    k1 = run("extract (no log), --host-ingest", fq, ["--host-ingest"], n)
    assert k0 == k1
run("extract -l -j, device ingest", fq, LOGS, n)
run("extract -l -j, --host-ingest", fq, LOGS + ["--host-ingest"], n)

if GZ:
    import zlib
    from concurrent.futures import ThreadPoolExecutor
    m = m_small
    raw = raw_small
    small = os.path.join(tmp, "e2e_small.fastq")
    open(small, "wb").write(raw)
    t0 = time.time()
    if m <= 4_000_000:
        subprocess.run(f"gzip -1 -c {small} > {small}.gz", shell=True, check=True)
    else:
        open(small + ".gz", "wb").close()

    def member(b):
        chunk = raw[b:b + 0xff00]
        co = zlib.compressobj(1, zlib.DEFLATED, -15)
        c = co.compress(chunk) + co.flush()
        return (bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0]) + (len(c) + 25).to_bytes(2, "little") + c +
                zlib.crc32(chunk).to_bytes(4, "little") + len(chunk).to_bytes(4, "little"))

    with ThreadPoolExecutor(16) as ex:
        parts = list(ex.map(member, range(0, len(raw), 0xff00)))
    bg = os.path.join(tmp, "e2e_small.bgzf.fastq.gz")
    open(bg, "wb").write(b"".join(parts) + bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
    print(f"compressed {m} reads ({len(raw) / 1e6:.0f} MB): gzip -1 {os.path.getsize(small + '.gz') / 1e6:.0f} MB, BGZF {os.path.getsize(bg) / 1e6:.0f} MB, in {time.time() - t0:.1f} s", flush=True)
    run(f"plain, {m} reads, device ingest", small, [], m)
    run(f"plain, {m} reads, --host-ingest", small, ["--host-ingest"], m)
    if m <= 4_000_000:
        run(f"gzip (one member), {m} reads, device ingest", small + ".gz", [], m)
        run(f"gzip (one member), {m} reads, --host-ingest", small + ".gz", ["--host-ingest"], m)
    for rep in range(2):
        run(f"BGZF, {m} reads, device ingest, members inflated on the device", bg, [], m)
        run(f"BGZF, {m} reads, device ingest, --host-codec (zlib on the host threads)", bg, ["--host-codec"], m)
    run(f"BGZF, {m} reads, --host-ingest, members inflated on the device", bg, ["--host-ingest"], m)
    run(f"BGZF, {m} reads, --host-ingest --host-codec", bg, ["--host-ingest", "--host-codec"], m)

#!/usr/bin/env python3
"""End-to-end wall time of the C++ CLI on a synthetic FASTQ (host parsing + PCIe + scan + write).
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
binp = os.path.join(ROOT, "merkurio_amd", "lib", "merkurio")
for label, extra in (("extract (no log)", []), ("extract -l -j", ["-l", os.path.join(tmp, "e2e.log"), "-j", os.path.join(tmp, "e2e.json")])):
    t0 = time.time()
    subprocess.run([binp, "extract", "-i", fq, "-f", km, "-o", os.path.join(tmp, "e2e_out"), *extra], check=True,
                   env=dict(os.environ, MERKURIO_TIMING="1"))
    dt = time.time() - t0
    kept = os.path.getsize(os.path.join(tmp, "e2e_out.fastq")) // (13 + 2 * L + 4)
    logs = sum(os.path.getsize(os.path.join(tmp, f)) for f in ("e2e.log", "e2e.json") if extra and os.path.exists(os.path.join(tmp, f)))
    print(f"{label}: {dt:.2f} s wall -> {n * L / dt / 1e9:.3f} Gbases/s end to end, {kept} reads extracted" + (f", {logs / 1e6:.0f} MB of logs" if extra else ""), flush=True)

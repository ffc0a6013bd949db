#!/usr/bin/env python3
"""End-to-end wall time of `merkurio tag` on synthetic SAM / BAM (host codec + PCIe + scan + write).
usage: tools/e2e_tag.py [n_records] [n_patterns] [one record in N carries a k-mer, default 100]"""
import os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
npat = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
every = int(sys.argv[3]) if len(sys.argv) > 3 else 100
L = 150
rng = np.random.default_rng(2)
tmp = os.environ.get("TMPDIR", "/tmp")
sam, km = os.path.join(tmp, "e2e.sam"), os.path.join(tmp, "e2e_kmers.txt")
t0 = time.time()
bases = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n, L))]
pats = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(npat, 31))]
if every == 1:  # every record carries a k-mer
    bases[:, 7:38] = pats[np.arange(n) % npat]
else:
    for i in range(0, n, every):  # 1 % of the records by default
        bases[i, 7:38] = pats[i % npat]
pre = np.array([f"r{i:010d}\t0\tchr1\t{i % 1000000 + 1:07d}\t60\t{L}M\t*\t0\t0\t" for i in range(n)], dtype="S41")
P = pre.dtype.itemsize
rec = np.empty((n, P + L + 1 + L + 1), dtype=np.uint8)
rec[:, :P] = pre.view(np.uint8).reshape(n, P)
rec[:, P:P + L] = bases
rec[:, P + L] = 9
rec[:, P + L + 1:P + 2 * L + 1] = ord("I")
rec[:, -1] = ord("\n")
with open(sam, "wb") as f:
    f.write(b"@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:chr1\tLN:2000000\n")
    rec.tofile(f)
open(km, "wb").write(b"\n".join(p.tobytes() for p in pats) + b"\n")
print(f"generated {n} records ({os.path.getsize(sam) / 1e6:.0f} MB SAM) in {time.time() - t0:.1f} s", flush=True)
binp = os.environ.get("MERKURIO_BIN") or os.path.join(ROOT, "merkurio_amd", "lib", "merkurio")
bam0, out = os.path.join(tmp, "e2e_in.bam"), os.path.join(tmp, "e2e_out")
# (the input BAM of the BAM rows is written with another tag name: records that already carry `km` take the reference's merge rule,
# which lives on the host path -- r05: BAM -> BAM keeps the records on the device, --host-ingest is the r04 path)
for label, args in (("SAM -> SAM", ["-i", sam, "-o", out + ".sam"]), ("SAM -> BAM", ["-i", sam, "-o", bam0, "-t", "zz"]),
                    ("BAM -> BAM, -m", ["-i", bam0, "-o", out + ".bam", "-m"]), ("BAM -> BAM, -m, --host-ingest", ["-i", bam0, "-o", out + "h.bam", "-m", "--host-ingest"]),
                    ("BAM -> BAM", ["-i", bam0, "-o", out + "3.bam"]), ("BAM -> BAM, --host-ingest", ["-i", bam0, "-o", out + "3h.bam", "--host-ingest"]),
                    ("BAM -> BAM (second run)", ["-i", bam0, "-o", out + "3.bam"]),
                    ("BAM -> SAM", ["-i", bam0, "-o", out + "2.sam"])):
    t0 = time.time()
    subprocess.run([binp, "tag", "-f", km, *args], check=True, env=dict(os.environ, MERKURIO_TIMING="1"))
    dt = time.time() - t0
    o = args[args.index("-o") + 1]
    print(f"{label}: {dt:.2f} s wall -> {n * L / dt / 1e9:.3f} Gbases/s end to end, output {os.path.getsize(o) / 1e6:.0f} MB", flush=True)

#!/usr/bin/env python3
"""The reference's own published benchmark shape, end to end (benchmarks/scripts/03-run-benchmarks.sh:94,113;
benchmarks/results/summary.md:16,24,35,46): `merkurio extract` on ~647 k x 101 bp FASTQ reads (the size of GAGE
S. aureus frag_1/frag_2.fastq, which the reference downloads; synthetic here), 1 and 100 31-mers sampled from the
reads (benchmarks/scripts/02-generate-kmers.sh:24-39), single-end and paired `-2 -r`.  Every run is a fresh
process (cold start: HIP runtime init, code-object load, file read, scan, write): RUNS runs, median / min wall,
next to the reference's published means (other hardware, one CPU core).
usage: tools/refbench.py [runs]"""
import os, statistics, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUNS = int(sys.argv[1]) if len(sys.argv) > 1 else 10
N, L = 647_052, 101
tmp = os.environ.get("TMPDIR", "/tmp")
rng = np.random.default_rng(20240)


def write_fastq(path, mate):
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(N, L))]
    hdr = np.array([f"@SRR022868.{i:07d}/{mate}\n" for i in range(N)], dtype="S21")
    H = hdr.dtype.itemsize
    assert H == len(f"@SRR022868.{0:07d}/{mate}\n")
    rec = np.empty((N, H + L + 3 + L + 1), dtype=np.uint8)
    rec[:, :H] = hdr.view(np.uint8).reshape(N, H)
    rec[:, H:H + L] = bases
    rec[:, H + L:H + L + 3] = np.frombuffer(b"\n+\n", dtype=np.uint8)
    rec[:, H + L + 3:H + 2 * L + 3] = ord("I")
    rec[:, -1] = ord("\n")
    rec.tofile(path)
    return bases


f1, f2 = os.path.join(tmp, "frag_1.fastq"), os.path.join(tmp, "frag_2.fastq")
b1 = write_fastq(f1, 1)
write_fastq(f2, 2)
# k-mers sampled from reads of file 1 at random offsets (the reference's recipe)
kfiles = {}
for n in (1, 100):
    rows = rng.choice(N, size=n, replace=False)
    offs = rng.integers(0, L - 31 + 1, size=n)
    p = os.path.join(tmp, f"{n}x31mers.txt")
    with open(p, "wb") as f:
        for r, o in zip(rows, offs):
            f.write(b1[r, o:o + 31].tobytes() + b"\n")
    kfiles[n] = p
binp = os.path.join(ROOT, "merkurio_amd", "lib", "merkurio")
out = os.path.join(tmp, "refbench_out")
print(f"# {N} reads x {L} bp per file ({os.path.getsize(f1) / 1e6:.0f} MB FASTQ), fresh process per run, {RUNS} runs each", flush=True)
CASES = [
    ("single, 1 x 31-mer (BNDMq)", 0.360, ["extract", "-i", f1, "-f", kfiles[1], "-o", out]),
    ("single, 100 x 31-mers (Aho-Corasick)", 0.513, ["extract", "-i", f1, "-f", kfiles[100], "-o", out]),
    ("paired -r, 1 x 31-mer (+RC)", 0.684, ["extract", "-1", f1, "-2", f2, "-r", "-f", kfiles[1], "-o", out]),
    ("paired -r, 100 x 31-mers (+RC)", 1.033, ["extract", "-1", f1, "-2", f2, "-r", "-f", kfiles[100], "-o", out]),
]
for label, ref_s, args in CASES:
    walls = []
    for i in range(RUNS + 2):  # two warm-ups (page cache), like hyperfine's warm-up runs
        t0 = time.perf_counter()
        subprocess.run([binp, *args], check=True, stdout=subprocess.DEVNULL)
        if i >= 2:
            walls.append(time.perf_counter() - t0)
    r = subprocess.run([binp, *args], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True,
                       env=dict(os.environ, MERKURIO_TIMING="1"))
    kept = sum(os.path.getsize(p) for p in (out + ".fastq", out + "_1.fastq", out + "_2.fastq") if os.path.exists(p) and
               os.path.getmtime(p) > time.time() - 5)
    print(f"{label}: median {statistics.median(walls):.3f} s, min {min(walls):.3f} s (reference, its own hardware, 1 core: "
          f"{ref_s:.3f} s mean); {kept} bytes extracted", flush=True)
    for line in r.stderr.splitlines():
        if line.startswith("[timing]"):
            print("    " + line, flush=True)

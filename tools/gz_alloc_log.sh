#!/bin/bash
# Runs on the GPU box: `merkurio extract` on a one-member .fastq.gz with the diagnostic build of the library that logs every
# device-buffer growth (python -m merkurio_amd.build --tag alloclog --flags "-DMK_ALLOC_LOG=1") next to the CLI's own timing lines.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
BIN=$ROOT/merkurio_amd/lib/merkurio
python3 - <<PY
import sys, zlib
sys.path.insert(0, "$ROOT"); sys.path.insert(0, "$ROOT/tests")
import numpy as np
from bench import _fastq_binned
d = _fastq_binned(4000000)
co = zlib.compressobj(1, zlib.DEFLATED, 31)
open("/tmp/al.fastq.gz", "wb").write(co.compress(d) + co.flush())
rng = np.random.default_rng(9)
pats = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(10000, 31))]
open("/tmp/al_kmers.txt", "wb").write(b"\n".join(p.tobytes() for p in pats) + b"\n")
PY
for rep in 1 2; do
  echo "== run $rep"
  LD_PRELOAD=$ROOT/merkurio_amd/lib/libmerkurio_hip_alloclog.so MERKURIO_TIMING=1 $BIN extract -f /tmp/al_kmers.txt -i /tmp/al.fastq.gz -o /tmp/al_out 2>&1 | grep "alloc\]\|timing"
done

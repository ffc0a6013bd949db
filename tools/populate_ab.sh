#!/bin/bash
# extract end to end with and without MAP_POPULATE on the input mapping (after tools/e2e_cli.py)
T=${TMPDIR:-/tmp}
for r in 1 2; do for pop in 1 0; do
  s=$(date +%s.%N)
  MERKURIO_MMAP_POPULATE=$pop MERKURIO_TIMING=1 merkurio_amd/lib/merkurio extract -i $T/e2e.fastq -f $T/e2e_kmers.txt -o $T/e2e_out 2> $T/bm.err
  e=$(date +%s.%N)
  echo "populate=$pop: $(python3 -c "print(round($e - $s, 3))") s wall; parse $(grep 'parse' $T/bm.err | awk '{print $(NF-1)}') s"
done; done

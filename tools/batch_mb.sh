#!/bin/bash
# end-to-end extract wall time against the CLI's batch size (after tools/e2e_cli.py left its files in $TMPDIR)
T=${TMPDIR:-/tmp}
for mb in 1024 512 256 128 64; do
  s=$(date +%s.%N)
  MERKURIO_TIMING=1 merkurio_amd/lib/merkurio extract -i $T/e2e.fastq -f $T/e2e_kmers.txt -o $T/e2e_out --batch-mb $mb 2> $T/bm.err
  e=$(date +%s.%N)
  echo "batch-mb $mb: $(python3 -c "print(round($e - $s, 3))") s wall; gather $(grep gather $T/bm.err | awk '{s+=$(NF-1)} END {print s}') s, scan $(grep 'H2D' $T/bm.err | awk '{s+=$(NF-1)} END {print s}') s, parse $(grep 'parse' $T/bm.err | awk '{print $(NF-1)}') s"
done

#!/bin/bash
# round 3, batch 8: 4-rank rehearsal of the driver's multi-GPU job at its default size on one device; 127-chunk
# tiles (halo chunk carried across four tiles) against the default; dense tag end to end
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
echo "# python bench.py --gpus 4 (default size: 100 M x 150 bp per rank) on ONE MI355X: rehearsal, ranks share the device, gloo" | tee gpurun_out/r03_rehearsal_n4.txt
( time timeout -k 10 500 python bench.py --gpus 4 --no-cpu-baseline ) 2>&1 | tail -6 | tee -a gpurun_out/r03_rehearsal_n4.txt
echo "# halo chunk carried across four tiles (127-chunk tiles, run 1) vs 31-chunk tiles in runs of 4 (default); Gbases/s, ms/step, scan kernel ms" | tee gpurun_out/r03_tile127.txt
for r in 1 2 3; do
  unset MERKURIO_LIB_PATH
  echo -n "default (31 x 4): " | tee -a gpurun_out/r03_tile127.txt; timeout -k 10 300 python bench.py --no-cpu-baseline --no-other-configs --steps 10 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms_avg'])" | tee -a gpurun_out/r03_tile127.txt
  export MERKURIO_LIB_PATH=$PWD/merkurio_amd/lib/libmerkurio_hip_t127.so
  echo -n "127-chunk tiles: " | tee -a gpurun_out/r03_tile127.txt; timeout -k 10 300 python bench.py --no-cpu-baseline --no-other-configs --steps 10 --tile-run 1 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms_avg'], j['summary']['hits'])" | tee -a gpurun_out/r03_tile127.txt
done
unset MERKURIO_LIB_PATH
export TMPDIR=/tmp
echo "# tools/e2e_tag.py 4000000 10000 1: EVERY record carries a k-mer" | tee gpurun_out/r03_e2e_tag_dense.txt
timeout -k 10 500 python tools/e2e_tag.py 4000000 10000 1 2>&1 | grep -v "^\[timing\]" | tee -a gpurun_out/r03_e2e_tag_dense.txt

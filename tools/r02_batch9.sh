#!/bin/bash
# r02 GPU batch 9: where does a verified occurrence's time go (headline workload, 1 % and 10 % of the reads hit)?
# ablation builds: 64 = q-gram hits not queued, 16 = level 3 dropped, 32 = level 3 without its flag store, 256 = no flag store
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
L=merkurio_amd/lib
one() { timeout -k 10 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; s=j['summary']; print(j['value'], 'Gbases/s kernel avg/min', r['kernel_ms_avg'], r['kernel_ms_min'], 'hits', s['hits'])"; }
{
for r in 1 2; do for pe in 100 10; do
  echo -n "plant_every=$pe full: "; one --steps 10 --plant-every $pe
  for a in 256 32 16 64; do echo -n "plant_every=$pe abl$a: "; MERKURIO_LIB_PATH=$L/libmerkurio_hip_abl$a.so one --steps 10 --plant-every $pe; done
done; echo -n "plant_every=0 full: "; one --steps 10 --plant-every 0; done
} > $O/r02_hit_breakdown.txt 2>&1; cat $O/r02_hit_breakdown.txt

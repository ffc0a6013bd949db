#!/bin/bash
# record lookup of verified occurrences: equal-length reads vs lengths uniform in [L - S, L + S] over the same bytes,
# with the coarse record index (mk_matcher_hint_record_lengths(m, 0)) and without (--no-rec-index: position quotient + gallop)
cd ${GRAFT_REPO_ROOT:-/root/repo}
one() { timeout -k 10 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; s=j['summary']; print(j['value'], 'Gbases/s ms/step', j['ms_per_step'], 'kernel avg/min', r['kernel_ms_avg'], r['kernel_ms_min'], 'frac', r['frac'], 'hits', s['hits'], 'records', s['records'])"; }
for pe in 100 10; do
  echo -n "plant_every=$pe equal lengths: "; one --steps 10 --plant-every $pe
  for sp in 1 60; do
    echo -n "plant_every=$pe ragged=$sp quotient + gallop: "; one --steps 10 --plant-every $pe --ragged $sp --no-rec-index
    echo -n "plant_every=$pe ragged=$sp coarse index:      "; one --steps 10 --plant-every $pe --ragged $sp
  done
done

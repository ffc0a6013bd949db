#!/bin/bash
# r02 GPU batch 12: four-rank rehearsals of bench.py on one GPU (gloo, ranks share the device): weak, strong, strong paired
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
{
for args in "--gpus 4 --records 2000000 --steps 3 --warmup 1 --patterns 2000" \
            "--gpus 4 --scaling strong --total-records 8000000 --steps 3 --warmup 1 --patterns 2000" \
            "--gpus 4 --scaling strong --paired --total-records 8000000 --steps 3 --warmup 1 --patterns 2000" \
            "--gpus 1 --scaling strong --total-records 8000000 --steps 3 --warmup 1 --patterns 2000 --no-cpu-baseline" \
            "--gpus 3 --scaling strong --total-records 8000000 --steps 3 --warmup 1 --patterns 2000"; do
  echo "== python bench.py $args"
  timeout -k 10 600 python bench.py $args 2>$O/r02_rehearsal.err | python -c "
import json,sys
lines=sys.stdin.read().splitlines()
for l in lines:
    if not l.startswith('{'): print('  [stdout]', l[:160])
j=json.loads([l for l in lines if l.startswith('{')][0]); print(' n_gpus', j['n_gpus'], 'scaling', j['scaling'], 'rehearsal', j.get('rehearsal'), 'value', j['value'], 'per-gpu records', j['config']['records_per_gpu'], 'reduce:', j['config']['counter_reduction'], 'summary', j['summary'])"
done
} > $O/r02_rehearsal.txt 2>&1; cat $O/r02_rehearsal.txt
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "out_of_memory or borders" 2>&1 | tail -3

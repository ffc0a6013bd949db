#!/bin/bash
# scan-kernel time against batch size (fixed launch/tail overhead vs streaming rate); headline pattern set and the C2 set
run() { echo -n "$*: "; timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 3 "$@" 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print(j['value'], 'Gbases/s kernel avg/min', r['kernel_ms_avg'], r['kernel_ms_min'], 'frac', r['frac'], j['config']['kernel'])"; }
for n in 1000000 2500000 5000000 10000000 20000000 50000000; do run --records $n --patterns 10000; done
for n in 1000000 10000000; do run --records $n --patterns 1024 --rc; done

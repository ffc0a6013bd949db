# throughput of the BASELINE.json configurations that fit one GPU (informative; the bench line is the headline)
run() { echo "== $*"; timeout -k 10 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(' ', j['value'], 'Gbases/s', j['ms_per_step'], 'ms/step kernel', j['roofline']['kernel_ms_avg'], 'frac', j['roofline']['frac'], j['config']['kernel'], j['config']['filter'], j['summary'])"; }
run --records 10000000 --patterns 1024 --rc --steps 20                      # C2
run --records 100000000 --patterns 10000 --steps 10                         # headline / C3 per-GPU shape
run --records 100000000 --patterns 10000 --steps 10 --mode hits             # with hit tuples (logging)
run --records 20000000 --patterns 10000 --steps 20 --mode hits              # C4 shape (tag needs hit sets)
run --records 12500000 --read-len 250 --patterns 500000 --k 21 --steps 5    # C5 per-GPU shard (100 M / 8)
run --records 100000000 --patterns 13 --steps 10                            # BNDMq domain (13 patterns)

#!/bin/bash
# quick look at the paired window pipeline's phases (tools/e2e_pairs.py at 4 M pairs, timing lines kept)
export TMPDIR=/tmp
python tools/e2e_pairs.py ${1:-4000000} 10000 200

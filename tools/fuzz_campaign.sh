#!/bin/bash
# long randomised differential campaign on the GPU box: tools/fuzz_campaign.sh [seeds] [base]
cd ${GRAFT_REPO_ROOT:-/root/repo}
S=${1:-300}; B=${2:-7000}
( time MERKURIO_FUZZ_SEEDS=$S MERKURIO_FUZZ_BASE=$B python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q ) > gpurun_out/fuzz_campaign.log 2>&1
tail -6 gpurun_out/fuzz_campaign.log

#!/bin/bash
# r02 GPU batch 26: dense (plain-load) variants for the runtime-q kernel families: full GPU suite
cd ${GRAFT_REPO_ROOT:-/root/repo}
( time timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=5 ) > gpurun_out/r02_gputest26.log 2>&1; rc=$?
tail -15 gpurun_out/r02_gputest26.log; exit $rc

#!/bin/bash
# round 5: whole GPU suite (durations) after the N-D work
set -u
OUT=gpurun_out/r05h
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > $OUT/tests_gpu.log 2>&1
echo "tests rc=$?" >> $OUT/tests_gpu.log
tail -30 $OUT/tests_gpu.log

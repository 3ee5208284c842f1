#!/bin/bash
# round 5: where the GPU suite's five minutes go (every test's duration)
set -u
OUT=gpurun_out/r05w
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
date +%s.%N > $OUT/t0
timeout 1500 python -m pytest tests -m gpu -x -q --durations=0 > $OUT/tests_gpu.log 2>&1
echo "tests rc=$?" >> $OUT/tests_gpu.log
date +%s.%N > $OUT/t1
tail -3 $OUT/tests_gpu.log

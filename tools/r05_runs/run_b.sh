#!/bin/bash
# round 5, second GPU call (default build of the library): the whole GPU suite with durations
set -u
OUT=gpurun_out/r05b
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1500 python -m pytest tests -m gpu -x -q --durations=60 > $OUT/tests_gpu.log 2>&1
echo "tests rc=$?" >> $OUT/tests_gpu.log
tail -80 $OUT/tests_gpu.log

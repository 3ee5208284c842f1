#!/bin/bash
set -u
OUT=gpurun_out/r05s
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_round5_gpu.py -x -q -k "work_groups" --durations=5 > $OUT/tests_nd2z.log 2>&1
echo "tests rc=$?" >> $OUT/tests_nd2z.log
tail -14 $OUT/tests_nd2z.log

#!/bin/bash
# round 5: copy-kernel variants (tools/membench6.hip) + the new host-thread test
set -u
OUT=gpurun_out/r05z
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 ./tools/membench6 > $OUT/membench6.log 2>&1; cat $OUT/membench6.log
timeout 600 python -m pytest tests/test_round5_gpu.py -m gpu -x -q -k "four_host_threads or captured_execute" > $OUT/threads.log 2>&1; tail -15 $OUT/threads.log

#!/bin/bash
# round 4, call e: fp64 2^21 / 2^22 on the persistent kernel; GPU suite; fp64 sweeps
set -u
OUT=gpurun_out/r04e
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python3 -m pytest tests/test_round4_gpu.py -q -m gpu -k "fused_long_fp64 or per_xcd" > $OUT/pytest_r4.log 2>&1; tail -25 $OUT/pytest_r4.log
S="python3 tools/fused_sweep.py"
timeout 900 $S \
  2097152 complex128 1 pipelined,auto,f:3:7,f:5:7,f:2:4 \
  2097152 complex128 4 pipelined,auto,f:3:7,f:5:7,f:2:4 \
  4194304 complex128 1 pipelined,auto,f:2:3 \
  4194304 complex128 4 pipelined,auto,f:2:3 \
  > $OUT/fp64_long.log 2>&1
cat $OUT/fp64_long.log
timeout 1500 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_all.log 2>&1; tail -5 $OUT/pytest_all.log

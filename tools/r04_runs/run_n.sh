#!/bin/bash
# round 4, call n: fp64 2^19 and the fp64 2-D shapes with a 512-point side on the persistent kernels
set -u
OUT=gpurun_out/r04n
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python3 -m pytest tests/test_round4_gpu.py -q -m gpu > $OUT/pytest_r4.log 2>&1; tail -5 $OUT/pytest_r4.log
S="python3 tools/fused_sweep.py"
timeout 900 $S \
  524288 complex128 1 pipelined,auto \
  524288 complex128 4 pipelined,auto,f:10:20,f:8:14 \
  512x512 complex128 1 pipelined,auto \
  512x512 complex128 4 pipelined,auto \
  512x1024 complex128 4 pipelined,auto \
  1024x512 complex128 4 pipelined,auto \
  1024x1024 complex128 4 auto \
  1048576 complex128 4 auto \
  > $OUT/sweep.log 2>&1
cat $OUT/sweep.log

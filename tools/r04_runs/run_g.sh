#!/bin/bash
# round 4, call g: the fp32 cube on 1024-thread work-groups (plane-sized items) against the two earlier forms
set -u
OUT=gpurun_out/r04g
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python3 -m pytest tests/test_round4_gpu.py -q -m gpu > $OUT/pytest_r4.log 2>&1; tail -8 $OUT/pytest_r4.log
S="python3 tools/fused_sweep.py"
timeout 900 $S \
  128x128x128 complex64 1 pipelined,auto,auto@MIFFT_PAIR=3,auto@MIFFT_PAIR=4 \
  128x128x128 complex64 4 pipelined,auto,auto@MIFFT_PAIR=3,auto@MIFFT_PAIR=4,f:4:8,f:5:10 \
  128x128x128 complex128 4 auto \
  > $OUT/cube_sweep.log 2>&1
cat $OUT/cube_sweep.log

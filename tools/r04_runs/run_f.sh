#!/bin/bash
# round 4, call f: wide against narrow tiles in the persistent two-pair kernel; PMC traffic of the cubes
set -u
OUT=gpurun_out/r04f
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python3 -m pytest tests/test_round4_gpu.py -q -m gpu -k "cube or fused_long or sequential" > $OUT/pytest_r4.log 2>&1; tail -8 $OUT/pytest_r4.log
S="python3 tools/fused_sweep.py"
timeout 900 $S \
  128x128x128 complex64 1 pipelined,auto,auto@MIFFT_PAIR=3 \
  128x128x128 complex64 4 pipelined,auto,auto@MIFFT_PAIR=3,f:4:8 \
  128x128x128 complex128 1 pipelined,auto,auto@MIFFT_PAIR=3 \
  128x128x128 complex128 4 pipelined,auto,auto@MIFFT_PAIR=3 \
  > $OUT/cube_sweep.log 2>&1
cat $OUT/cube_sweep.log
timeout 1200 python3 tools/pmc_traffic.py --tag r04 cube cubed > $OUT/pmc_cube.log 2>&1; cat $OUT/pmc_cube.log | tail -5
cp profiles/traffic_cube*.json profiles/r04_cube*_kernel_stats.csv $OUT/ 2>/dev/null

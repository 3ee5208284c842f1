#!/bin/bash
# round 4, call s: 2-D shapes with a 512-point axis on 32-column tiles for that axis' pass
set -u
OUT=gpurun_out/r04s
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python3 -m pytest tests/test_errors_gpu.py tests/test_round4_gpu.py -q -m gpu -k "fused_2d or rectangles or squares" > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
S="python3 tools/fused_sweep.py"
timeout 900 $S \
  512x512 complex64 1 auto,auto@MIFFT_NARROW_TILES=1 \
  512x512 complex64 4 auto,auto@MIFFT_NARROW_TILES=1 \
  512x1024 complex64 4 auto,auto@MIFFT_NARROW_TILES=1 \
  1024x512 complex64 4 auto,auto@MIFFT_NARROW_TILES=1 \
  1024x512 complex64 1 auto,auto@MIFFT_NARROW_TILES=1 \
  > $OUT/sweep.log 2>&1
cat $OUT/sweep.log

#!/bin/bash
# round 4, call o: fp32 2^16 ... 2^18 on 32-column tiles (16-byte lanes, 256-byte segments) against the 16-column ones
set -u
OUT=gpurun_out/r04o
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python3 -m pytest tests/test_errors_gpu.py -q -m gpu -k "fused_two_pass_kernel" > $OUT/pytest.log 2>&1; tail -12 $OUT/pytest.log
S="python3 tools/fused_sweep.py"
timeout 900 $S \
  262144 complex64 1 auto,auto@MIFFT_NARROW_TILES=1,f:56:112 \
  262144 complex64 4 auto,auto@MIFFT_NARROW_TILES=1,f:56:112,f:28:56 \
  131072 complex64 1 auto,f:112:224,f:56:112,f:112:224@MIFFT_NARROW_TILES=1 \
  131072 complex64 4 auto,f:112:224,f:56:112,f:56:112@MIFFT_NARROW_TILES=1 \
  65536 complex64 1 auto,f:112:224,f:56:112 \
  65536 complex64 4 auto,f:112:224,f:224:448,f:56:112,f:56:112@MIFFT_NARROW_TILES=1 \
  > $OUT/sweep.log 2>&1
cat $OUT/sweep.log

#!/bin/bash
# round 4, call q: 2^19 with a 32-column second pass; the chain threshold against the wide-tile persistent kernel at 128 / 256 MiB
set -u
OUT=gpurun_out/r04q
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python3 -m pytest tests/test_errors_gpu.py tests/test_round4_gpu.py -q -m gpu -k "fused_two_pass_kernel or ring_rule or wide_tiles" > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
S="python3 tools/fused_sweep.py"
timeout 900 $S \
  524288 complex64 1 auto,auto@MIFFT_NARROW_TILES=1 \
  524288 complex64 4 auto,auto@MIFFT_NARROW_TILES=1 \
  65536 complex64 0.125 auto,f:32:64,f:64:128 \
  65536 complex64 0.25 auto,f:64:128,f:112:224 \
  262144 complex64 0.125 auto,f:16:32 \
  262144 complex64 0.25 auto,f:28:56,f:16:32 \
  1048576 complex64 0.25 auto,f:8:16 \
  65536 complex128 0.25 auto,f:28:56 \
  > $OUT/sweep.log 2>&1
cat $OUT/sweep.log

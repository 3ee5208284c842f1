#!/bin/bash
# round 5: fp64 1024-point strided passes on 8-column tiles (256 threads, two work-groups per CU) against the 16-column 512-thread tiles, plain launches
set -u
OUT=gpurun_out/r05x
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python tools/fused_sweep.py \
  1048576 complex128 1 chain,chain@MIFFT_NARROW_TILES=3,pipelined,pipelined@MIFFT_NARROW_TILES=3,auto \
  524288 complex128 1 chain,chain@MIFFT_NARROW_TILES=3,pipelined,pipelined@MIFFT_NARROW_TILES=3,auto \
  1024x1024 complex128 1 chain,chain@MIFFT_NARROW_TILES=3,pipelined,pipelined@MIFFT_NARROW_TILES=3,auto \
  1048576 complex128 0.125 chain,chain@MIFFT_NARROW_TILES=3 \
  > $OUT/col3_w8_f64_plain.log 2>&1
cat $OUT/col3_w8_f64_plain.log

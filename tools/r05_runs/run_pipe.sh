#!/bin/bash
# round 5: chunk size of the pipelined strategy on the shapes that have no persistent kernel (1 GiB per side)
set -u
OUT=gpurun_out/r05pipe
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
V=pipelined@PYFFT_AMD_PIPE_MB=16,pipelined@PYFFT_AMD_PIPE_MB=32,pipelined,pipelined@PYFFT_AMD_PIPE_MB=96,pipelined@PYFFT_AMD_PIPE_MB=128,chain
timeout 1200 python tools/fused_sweep.py \
  32x32x128 complex64 1 $V \
  32x32x128 complex128 1 $V \
  256x128x128 complex64 1 $V \
  256x4096 complex64 1 $V \
  4096x256 complex64 1 $V \
  8388608 complex64 1 $V \
  32768 complex128 1 $V \
  2048x2048 complex128 1 $V \
  > $OUT/pipe_chunk_sweep.log 2>&1
cat $OUT/pipe_chunk_sweep.log

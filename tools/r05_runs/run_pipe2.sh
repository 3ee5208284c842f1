#!/bin/bash
# round 5: number of side streams of the pipelined strategy (default 2) on the shapes that have no persistent kernel (1 GiB per side)
set -u
OUT=gpurun_out/r05pipe
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
V=pipelined,pipelined@PYFFT_AMD_PIPE_STREAMS=3,pipelined@PYFFT_AMD_PIPE_STREAMS=4,pipelined@PYFFT_AMD_PIPE_STREAMS=3@PYFFT_AMD_PIPE_MB=48,pipelined@PYFFT_AMD_PIPE_STREAMS=4@PYFFT_AMD_PIPE_MB=32,pipelined@PYFFT_AMD_PIPE_STREAMS=1
timeout 1200 python tools/fused_sweep.py \
  32x32x128 complex64 1 $V \
  256x128x128 complex64 1 $V \
  256x4096 complex64 1 $V \
  4096x256 complex64 1 $V \
  8388608 complex64 1 $V \
  32768 complex128 1 $V \
  256x256x256 complex128 4 $V \
  > $OUT/pipe_streams_sweep.log 2>&1
cat $OUT/pipe_streams_sweep.log

#!/bin/bash
# round 4, call r: plain 32-column kernels in the chain / pipelined passes of L = 256 / 512 (fp32): GPU suite, then A/B against 16 columns
set -u
OUT=gpurun_out/r04r
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1800 python3 -m pytest tests -q -m gpu -x > $OUT/pytest_all.log 2>&1; tail -6 $OUT/pytest_all.log
S="python3 tools/fused_sweep.py"
timeout 900 $S \
  65536 complex64 0.125 auto,auto@MIFFT_NARROW_TILES=1 \
  65536 complex64 0.25 auto,auto@MIFFT_NARROW_TILES=1 \
  131072 complex64 0.25 auto,auto@MIFFT_NARROW_TILES=1 \
  262144 complex64 0.125 auto,auto@MIFFT_NARROW_TILES=1 \
  262144 complex64 0.25 auto,auto@MIFFT_NARROW_TILES=1 \
  65536 complex64 2 pipelined,pipelined@MIFFT_NARROW_TILES=1 \
  262144 complex64 2 pipelined,pipelined@MIFFT_NARROW_TILES=1 \
  256x256 complex64 0.25 auto,auto@MIFFT_NARROW_TILES=1 \
  256x256 complex64 2 auto,auto@MIFFT_NARROW_TILES=1 \
  512x512 complex64 0.25 auto,auto@MIFFT_NARROW_TILES=1 \
  256x1024 complex64 2 auto,auto@MIFFT_NARROW_TILES=1 \
  65536 complex64 0.03125 auto,auto@MIFFT_NARROW_TILES=1 \
  > $OUT/sweep.log 2>&1
cat $OUT/sweep.log

#!/bin/bash
# round 4, call m: full GPU suite on the tree with the radix-32 XY tile (plain and persistent), anchored twiddles in colx, fp64 2^16 fused
set -u
OUT=gpurun_out/r04m
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1800 python3 -m pytest tests -q -m gpu > $OUT/pytest_all.log 2>&1; tail -6 $OUT/pytest_all.log
S="python3 tools/fused_sweep.py"
timeout 900 $S \
  128x128x128 complex64 0.03125 auto \
  128x128x128 complex64 0.25 auto \
  128x128x128 complex64 1 pipelined,auto,auto@MIFFT_PAIR=3 \
  128x128x128 complex64 4 pipelined,auto,auto@MIFFT_PAIR=3 \
  2097152 complex128 4 pipelined,auto \
  4194304 complex128 4 auto \
  65536 complex128 1 auto \
  > $OUT/sweep.log 2>&1
cat $OUT/sweep.log
timeout 300 python3 tools/small_batch_probe.py sp > $OUT/small_batch.log 2>&1
timeout 300 python3 tools/small_batch_probe.py dp >> $OUT/small_batch.log 2>&1
cut -c1-100 $OUT/small_batch.log

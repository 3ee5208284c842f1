#!/bin/bash
# round 5, third GPU call: pair chains (tests + before / after), then the whole GPU suite with durations
set -u
OUT=gpurun_out/r05c
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_round5_gpu.py -x -q -k "pair_chains" > $OUT/tests_pairs.log 2>&1
echo "tests rc=$?" >> $OUT/tests_pairs.log
tail -15 $OUT/tests_pairs.log
# before (MIFFT_PAIR=1: no pass pairs, the round-4 chains) / after at 2 GiB per side and at the reference's 32 MiB
for v in 1 0; do
  echo "MIFFT_PAIR=$v" >> $OUT/pair_chains.log
  MIFFT_PAIR=$v timeout 600 python tools/fused_sweep.py 4096x256 complex64 2 auto 4096x512 complex64 2 auto 4096x1024 complex64 2 auto 4096x2048 complex64 2 auto 4096x4096 complex64 2 auto \
     32x32x2048 complex64 2 auto 32x32x4096 complex64 2 auto 16x16x2048 complex64 2 auto 4096x256 complex128 2 auto 4096x1024 complex128 2 auto 32x32x1024 complex128 2 auto 32x32x2048 complex128 2 auto \
     4096x256 complex64 0.03125 auto 32x32x2048 complex64 0.03125 auto >> $OUT/pair_chains.log 2>&1
done
cat $OUT/pair_chains.log
timeout 1500 python -m pytest tests -m gpu -x -q --durations=40 > $OUT/tests_gpu.log 2>&1
echo "tests rc=$?" >> $OUT/tests_gpu.log
tail -60 $OUT/tests_gpu.log

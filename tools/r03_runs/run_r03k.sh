#!/bin/bash
set -u
OUT=gpurun_out/r03k
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
for pair in 0 1; do for dt in complex64 complex128; do for b in 64 2; do
  [ $dt = complex128 ] && bb=$((b/2)) || bb=$b
  [ $bb -lt 1 ] && bb=1
  MIFFT_PAIR=$pair python3 tools/quick_bench.py one 128x128x128 $dt $bb | tail -1 | sed "s/^/pair=$pair /"
done; done; done > $OUT/cube128.log 2>&1
cat $OUT/cube128.log | sed 's/passes=\[.*\]//'
timeout 600 python3 -m pytest tests/test_errors_gpu.py tests/test_random_sweep_gpu.py -x -q -m gpu -k "128" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log

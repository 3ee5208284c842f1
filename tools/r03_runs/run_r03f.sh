#!/bin/bash
set -u
OUT=gpurun_out/r03f
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python3 -m pytest tests/test_round3_gpu.py -x -q -m gpu -k "l2048 or pass_pairs" > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
timeout 900 python3 tools/quick_bench.py f64 > $OUT/f64.log 2>&1
cat $OUT/f64.log | sed 's/passes=\[.*\]//'
timeout 600 python3 tools/quick_bench.py one 2097152 complex128 256 >> $OUT/f64.log 2>&1
timeout 600 python3 tools/quick_bench.py one 2048x2048 complex128 64 >> $OUT/f64.log 2>&1
tail -2 $OUT/f64.log | sed 's/passes=\[.*\]//'

#!/bin/bash
set -u
OUT=gpurun_out/r03l
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python3 -m pytest tests/test_round3_gpu.py -x -q -m gpu -k "register_only" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
timeout 900 python3 -m pytest tests/test_errors_gpu.py tests/test_random_sweep_gpu.py -x -q -m gpu > $OUT/pytest2.log 2>&1; tail -3 $OUT/pytest2.log
timeout 600 python3 tools/quick_bench.py 3d > $OUT/3d.log 2>&1; sed 's/passes=\[.*\]//' $OUT/3d.log
for dt in complex64 complex128; do for sh in 16x16x128 32x32x128 128x128x128; do python3 tools/quick_bench.py one $sh $dt 256 | tail -1; done; done 2>&1 | sed 's/passes=\[.*\]//' | tee $OUT/table_rows.log
timeout 300 python3 tools/small_batch_probe.py sp > $OUT/small.log 2>&1; timeout 300 python3 tools/small_batch_probe.py dp >> $OUT/small.log 2>&1; cut -c1-110 $OUT/small.log

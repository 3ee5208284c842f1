#!/bin/bash
# round 4, call h: the single-buffer N-D smooth kernel; full GPU suite on the final tree; the size logs with long enough timing blocks
set -u
OUT=gpurun_out/r04h
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python3 -m pytest tests/test_round4_gpu.py -q -m gpu -k "smooth or bluestein or generic" > $OUT/pytest_r4.log 2>&1; tail -8 $OUT/pytest_r4.log
timeout 600 python3 tools/mixed_probe.py > $OUT/mixed.log 2>&1; tail -16 $OUT/mixed.log
timeout 1500 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_all.log 2>&1; tail -5 $OUT/pytest_all.log
timeout 900 python3 tools/quick_bench.py 1d > $OUT/long_1d.log 2>&1
timeout 900 python3 tools/quick_bench.py f64 >> $OUT/long_1d.log 2>&1
echo "# the two-pass fp32 sizes at 1 GiB and at 8 GiB per side" >> $OUT/long_1d.log
timeout 900 python3 tools/quick_bench.py 1d1g >> $OUT/long_1d.log 2>&1
timeout 900 python3 tools/quick_bench.py 1d8g >> $OUT/long_1d.log 2>&1
timeout 900 python3 tools/quick_bench.py r4 > $OUT/r4_shapes.log 2>&1
sed 's/passes=\[.*\]//' $OUT/long_1d.log | tail -16

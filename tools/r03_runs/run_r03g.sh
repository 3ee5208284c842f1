#!/bin/bash
set -u
OUT=gpurun_out/r03g
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1200 python3 -m pytest tests/test_round3_gpu.py tests/test_round2_gpu.py -x -q -m gpu -k "tiled" > $OUT/pytest_tiled.log 2>&1
tail -4 $OUT/pytest_tiled.log
timeout 600 python3 tools/generic_probe.py > $OUT/generic.log 2>&1
cat $OUT/generic.log
timeout 1500 python3 tools/fusedx_probe.py > $OUT/fusedx.log 2>&1
cat $OUT/fusedx.log

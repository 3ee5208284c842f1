#!/bin/bash
set -u
OUT=gpurun_out/r03n
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python3 -m pytest tests/test_round3_gpu.py tests/test_round2_gpu.py -x -q -m gpu -k "mixed or any_size or tiled" > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
timeout 600 python3 tools/generic_probe.py > $OUT/generic.log 2>&1; cat $OUT/generic.log

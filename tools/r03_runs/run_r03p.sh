#!/bin/bash
# round 3, step p: mixed-radix kernel with composite radices and register edges -- parity subset, generic_probe, mixed_probe
set -u
OUT=gpurun_out/r03p
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python3 -m pytest tests/test_round3_gpu.py tests/test_round2_gpu.py -x -q -m gpu -k "mixed or any_size or tiled or direct or generic" > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
timeout 600 python3 tools/generic_probe.py > $OUT/generic.log 2>&1; cat $OUT/generic.log
timeout 600 python3 tools/mixed_probe.py > $OUT/mixed.log 2>&1; cat $OUT/mixed.log

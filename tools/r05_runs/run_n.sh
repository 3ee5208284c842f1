#!/bin/bash
set -u
OUT=gpurun_out/r05n
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1200 python -m pytest tests -m gpu -x -q -k "fixed_shape or tiled or work_groups or nd or reference_error_grid or random" > $OUT/tests_nd.log 2>&1
echo "tests rc=$?" >> $OUT/tests_nd.log
tail -8 $OUT/tests_nd.log

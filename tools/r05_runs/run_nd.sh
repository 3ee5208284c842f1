#!/bin/bash
# round 5: the routed tiny N-D shapes -- their tests (with the soak cases), then the whole default suite
set -u
OUT=gpurun_out/r05nd2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
PYFFT_AMD_SWEEP=1 timeout 900 python -m pytest tests/test_round5_gpu.py -m gpu -x -q -k "tiny_nd" --durations=10 > $OUT/tiny_tests.log 2>&1; tail -14 $OUT/tiny_tests.log
timeout 1500 python -m pytest tests -m gpu -x -q --durations=6 > $OUT/tests_gpu.log 2>&1
echo "tests rc=$?" >> $OUT/tests_gpu.log
tail -10 $OUT/tests_gpu.log

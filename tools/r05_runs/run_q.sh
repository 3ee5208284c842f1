#!/bin/bash
# round 5: seeded random soak through the public API (tests/test_random_sweep_gpu.py) after the N-D work, and the soak-switch cases of the suite
set -u
OUT=gpurun_out/r05q
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
PYFFT_AMD_SWEEP=2500:20261005:21 timeout 1500 python -m pytest tests/test_random_sweep_gpu.py -x -q > $OUT/soak_random.log 2>&1
echo "rc=$?" >> $OUT/soak_random.log
tail -4 $OUT/soak_random.log
PYFFT_AMD_SWEEP=60:20261002 timeout 1500 python -m pytest tests -m gpu -x -q -k "not random_case" > $OUT/soak_suite.log 2>&1
echo "rc=$?" >> $OUT/soak_suite.log
tail -4 $OUT/soak_suite.log

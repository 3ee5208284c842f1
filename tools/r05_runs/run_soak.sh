#!/bin/bash
# round 5, final tree: the whole GPU suite WITH the soak cases (PYFFT_AMD_SWEEP=1), then the persistent-kernel soak
set -u
OUT=gpurun_out/r05soak
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
PYFFT_AMD_SWEEP=1 timeout 3000 python -m pytest tests -m gpu -q --durations=8 > $OUT/tests_gpu_sweep.log 2>&1
echo "tests rc=$?" >> $OUT/tests_gpu_sweep.log
tail -14 $OUT/tests_gpu_sweep.log
timeout 1200 python tools/persistent_soak.py > $OUT/persistent_soak.log 2>&1; tail -4 $OUT/persistent_soak.log

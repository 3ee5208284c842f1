#!/bin/bash
# round 5: whole GPU suite on the default library, then the development library's own tests (PYFFT_AMD_DEV_BUILD=1 -> libmifft_dev.so)
set -u
OUT=gpurun_out/r05v2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 > $OUT/tests_gpu.log 2>&1
echo "tests rc=$?" >> $OUT/tests_gpu.log
tail -22 $OUT/tests_gpu.log
PYFFT_AMD_DEV_BUILD=1 timeout 900 python -m pytest tests -m gpu -x -q -k "xcd2 or per_xcd or sequential or wide_tiles or alternating_counter or fused_2d_split_row_first or direct_abi" > $OUT/tests_gpu_dev.log 2>&1
echo "tests rc=$?" >> $OUT/tests_gpu_dev.log
tail -6 $OUT/tests_gpu_dev.log

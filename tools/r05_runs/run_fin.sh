#!/bin/bash
# round 5, after the write-through work: whole GPU suite (default library), development library's tests, two quick A/B rows
set -u
OUT=gpurun_out/r05fin
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > $OUT/tests_gpu.log 2>&1
echo "tests rc=$?" >> $OUT/tests_gpu.log
tail -14 $OUT/tests_gpu.log
PYFFT_AMD_DEV_BUILD=1 timeout 900 python -m pytest tests -m gpu -x -q -k "xcd2 or per_xcd or sequential or wide_tiles or alternating_counter or fused_2d_split_row_first or direct_abi" > $OUT/tests_gpu_dev.log 2>&1
echo "tests rc=$?" >> $OUT/tests_gpu_dev.log
tail -3 $OUT/tests_gpu_dev.log
timeout 300 python tools/fused_sweep.py 32x32 float32 0.03125 auto,auto@PYFFT_AMD_NO_ND_GENERIC=1 32x32 float32 1 auto,auto@PYFFT_AMD_NO_ND_GENERIC=1 32x32x32 float32 1 auto,auto@PYFFT_AMD_NO_ND_GENERIC=1 2>&1 | cut -c1-150 | tee $OUT/split_32x32.log
timeout 300 python tools/small_batch_probe.py dp 2>&1 | cut -c1-160 | tee $OUT/small_batch_dp.log

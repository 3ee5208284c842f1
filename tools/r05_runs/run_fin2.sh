#!/bin/bash
# round 5, after the late pass pairs: the pair tests with the soak cases, the whole GPU suite (default library), the development library's tests
set -u
OUT=gpurun_out/r05fin2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
PYFFT_AMD_SWEEP=1 timeout 900 python -m pytest tests/test_round5_gpu.py -m gpu -x -q -k "pass_pairs_for_256" 2>&1 | tail -3
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > $OUT/tests_gpu.log 2>&1
echo "tests rc=$?" >> $OUT/tests_gpu.log
tail -14 $OUT/tests_gpu.log
PYFFT_AMD_DEV_BUILD=1 timeout 900 python -m pytest tests -m gpu -x -q -k "xcd2 or per_xcd or sequential or wide_tiles or alternating_counter or fused_2d_split_row_first or direct_abi" > $OUT/tests_gpu_dev.log 2>&1
echo "tests rc=$?" >> $OUT/tests_gpu_dev.log
tail -3 $OUT/tests_gpu_dev.log
V=auto,auto@MIFFT_PAIR=1
timeout 600 python tools/fused_sweep.py 64x256x256 complex64 1 $V 128x256x256 complex64 1 $V 64x256x256 complex64 0.03125 $V 128x128x256 complex64 1 auto 2>&1 | cut -c1-150 | tee $OUT/late_pairs_f32.log

#!/bin/bash
# round 5: pass pairs for the fp64 shapes (z, y, 256): tests, then the default against MIFFT_PAIR=1 (the three-launch chains) at 1 GiB and 4 GiB
set -u
OUT=gpurun_out/r05lp
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
PYFFT_AMD_SWEEP=1 timeout 900 python -m pytest tests/test_round5_gpu.py -m gpu -x -q -k "pass_pairs_for_256" --durations=6 2>&1 | tail -12
V=auto,auto@MIFFT_PAIR=1
timeout 1200 python tools/fused_sweep.py 128x256x64 complex128 1 $V 64x256x64 complex128 1 $V 32x256x64 complex128 1 $V 256x256x64 complex128 1 $V 32x256x256 complex64 1 $V 64x256x64 complex128 0.03125 $V 32x256x256 complex64 0.03125 $V 2>&1 | cut -c1-150 | tee $OUT/late_pairs3.log

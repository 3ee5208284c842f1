#!/bin/bash
# round 5: A/B of the one-tile-per-CU fp32 N-D shapes (32768 points): 512 threads x 64 points against 1024 threads x 32 points
set -u
OUT=gpurun_out/r05d
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
python - > $OUT/nd2_huge_ab.log 2>&1 <<'PY'
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tools")
import numpy
from pyfft_amd import _native as N
import fused_sweep as fs
shapes = [(16, 16, 128), (32, 32, 32), (8, 64, 64), (128, 256), (256, 128), (32, 1024), (1024, 32), (512, 64)]
for gib in (0.03125, 1.0):
    for alt in (0, 4):
        N.lib.mifft_debug_set(N.DEBUG_ALT_ROWS, alt)
        print("# %s GiB per side, MIFFT_DEBUG_ALT_ROWS = %d (%s)" % (gib, alt, "1024 threads x 32 points" if alt else "512 threads x 64 points"), flush=True)
        for sh in shapes:
            fs.sweep(sh, "complex64", gib, ["auto"], reps=5, iters=10)
PY
cat $OUT/nd2_huge_ab.log
# pass-pair chains: before (MIFFT_PAIR=1: the round-4 chains, three launches) / after, 1 GiB per side (the tail survey's size) and 32 MiB
timeout 900 python tools/fused_sweep.py 4096x256 complex64 1 auto@MIFFT_PAIR=1,auto 4096x512 complex64 1 auto@MIFFT_PAIR=1,auto 4096x1024 complex64 1 auto@MIFFT_PAIR=1,auto \
   4096x2048 complex64 1 auto@MIFFT_PAIR=1,auto 4096x4096 complex64 1 auto@MIFFT_PAIR=1,auto 32x32x2048 complex64 1 auto@MIFFT_PAIR=1,auto 32x32x4096 complex64 1 auto@MIFFT_PAIR=1,auto \
   16x16x2048 complex64 1 auto@MIFFT_PAIR=1,auto 16x16x4096 complex64 1 auto@MIFFT_PAIR=1,auto 4096x256 complex128 1 auto@MIFFT_PAIR=1,auto 4096x512 complex128 1 auto@MIFFT_PAIR=1,auto \
   4096x1024 complex128 1 auto@MIFFT_PAIR=1,auto 4096x2048 complex128 1 auto@MIFFT_PAIR=1,auto 32x32x1024 complex128 1 auto@MIFFT_PAIR=1,auto 32x32x2048 complex128 1 auto@MIFFT_PAIR=1,auto \
   16x16x1024 complex128 1 auto@MIFFT_PAIR=1,auto 4096x256 complex64 0.03125 auto@MIFFT_PAIR=1,auto 32x32x2048 complex64 0.03125 auto@MIFFT_PAIR=1,auto > $OUT/pair_chains_ab.log 2>&1
cat $OUT/pair_chains_ab.log
timeout 1500 python -m pytest tests -m gpu -x -q --durations=0 > $OUT/tests_gpu_durations.log 2>&1
echo "tests rc=$?" >> $OUT/tests_gpu_durations.log
tail -4 $OUT/tests_gpu_durations.log

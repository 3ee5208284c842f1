#!/bin/bash
# round 5 experiment: the two-per-CU N-D shapes (16384 points fp32 / 8192 fp64) split in two as well
set -u
OUT=gpurun_out/r05f
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
python - > $OUT/nd2z_big_ab.log 2>&1 <<'PY'
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tools")
import numpy
from pyfft_amd import _native as N
import fused_sweep as fs
S32 = [(128, 128), (256, 64), (64, 256), (16, 32, 32), (32, 32, 16), (16, 1024)]
S64 = [(64, 128), (128, 64), (16, 16, 32), (32, 16, 16)]
for gib in (0.03125, 0.25, 1.0):
    for alt in (6, 0):
        N.lib.mifft_debug_set(N.DEBUG_ALT_ROWS, alt)
        print("# %s GiB per side, %s" % (gib, "two work-groups per transform (fft_nd2z)" if alt == 0 else "one tile per transform (fft_nd2 big)"), flush=True)
        for sh in S32:
            fs.sweep(sh, "complex64", gib, ["auto"], reps=5, iters=10)
        for sh in S64:
            fs.sweep(sh, "complex128", gib, ["auto"], reps=5, iters=10)
PY
cut -c1-150 $OUT/nd2z_big_ab.log

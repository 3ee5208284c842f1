#!/bin/bash
set -u
OUT=gpurun_out/r05p
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
python - > $OUT/nd2z_small4_ab.log 2>&1 <<'PY'
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tools")
import numpy
from pyfft_amd import _native as N
import fused_sweep as fs
S = [(16, 16, 128), (128, 256), (256, 128), (512, 64), (8, 64, 64)]
for gib in (0.03125, 0.125):
    for alt in (0, 8):
        N.lib.mifft_debug_set(N.DEBUG_ALT_ROWS, alt)
        print("# %s GiB per side, %s work-groups per transform" % (gib, "FOUR" if alt == 8 else "two"), flush=True)
        for sh in S:
            fs.sweep(sh, "complex64", gib, ["auto"], reps=5, iters=10)
PY
cut -c1-150 $OUT/nd2z_small4_ab.log

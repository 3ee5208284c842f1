#!/bin/bash
set -u
OUT=gpurun_out/r05l
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
python - > $OUT/nd2z_radix32_ab.log 2>&1 <<'PY'
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tools")
import numpy
from pyfft_amd import _native as N
import fused_sweep as fs
S = [(1024, 32), (32, 1024), (512, 64), (256, 128), (128, 256), (16, 16, 128)]
for gib in (0.03125, 1.0):
    for alt in (0, 7):
        N.lib.mifft_debug_set(N.DEBUG_ALT_ROWS, alt)
        print("# %s GiB per side, two work-groups per transform, %s" % (gib, "radix-32 stage lists" if alt == 7 else "automatic stage lists (radix <= 16)"), flush=True)
        for sh in S:
            fs.sweep(sh, "complex64", gib, ["auto"], reps=5, iters=10)
PY
cut -c1-150 $OUT/nd2z_radix32_ab.log

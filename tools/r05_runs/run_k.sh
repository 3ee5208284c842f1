#!/bin/bash
# round 5: 32^3 with radix-32 stages on two work-groups per transform; (32, 32, 128) on eight
set -u
OUT=gpurun_out/r05k
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
python - > $OUT/nd2z_more_ab.log 2>&1 <<'PY'
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tools")
import numpy
from pyfft_amd import _native as N
import fused_sweep as fs
for gib in (0.03125, 0.25, 1.0):
    for alt in (6, 0):
        N.lib.mifft_debug_set(N.DEBUG_ALT_ROWS, alt)
        print("# %s GiB per side, %s" % (gib, "several work-groups per transform (fft_nd2z)" if alt == 0 else "fft_nd2z off (MIFFT_DEBUG_ALT_ROWS = 6)"), flush=True)
        fs.sweep((32, 32, 32), "complex64", gib, ["auto"], reps=5, iters=10)
        fs.sweep((32, 32, 128), "complex64", gib, ["auto"], reps=5, iters=10)
PY
cut -c1-150 $OUT/nd2z_more_ab.log

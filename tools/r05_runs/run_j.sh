#!/bin/bash
# round 5 experiment: the fp64 one-tile shapes whose halves spill, as FOUR quarters of 4096 points
set -u
OUT=gpurun_out/r05j
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
python - > $OUT/nd2z_f64_quarters_ab.log 2>&1 <<'PY'
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tools")
import numpy
from pyfft_amd import _native as N
import fused_sweep as fs
for gib in (0.03125, 0.25, 1.0):
    for alt in (6, 0):
        N.lib.mifft_debug_set(N.DEBUG_ALT_ROWS, alt)
        print("# %s GiB per side, %s" % (gib, "four work-groups per transform (fft_nd2z)" if alt == 0 else "one tile per CU (fft_nd2 huge)"), flush=True)
        for sh in ((128, 128), (64, 16, 16)):
            fs.sweep(sh, "complex128", gib, ["auto"], reps=5, iters=10)
PY
cut -c1-150 $OUT/nd2z_f64_quarters_ab.log

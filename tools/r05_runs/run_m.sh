#!/bin/bash
set -u
OUT=gpurun_out/r05m
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
python - > $OUT/nd2_radix32_ab.log 2>&1 <<'PY'
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tools")
import numpy
from pyfft_amd import _native as N
import fused_sweep as fs
S = [(32, 512), (16, 32, 32), (32, 32, 16), (32, 16, 32), (8, 32, 64), (32, 8, 64), (32, 64, 8), (4, 32, 128), (32, 4, 128), (32, 128, 4), (16, 32, 32)]
for gib in (1.0, 0.03125):
    for alt in (0, 7):
        N.lib.mifft_debug_set(N.DEBUG_ALT_ROWS, alt)
        print("# %s GiB per side, two-per-CU tiles, %s" % (gib, "32-point axes as one radix-32 stage" if alt == 7 else "automatic stage lists (radix <= 16)"), flush=True)
        for sh in S:
            fs.sweep(sh, "complex64", gib, ["auto"], reps=5, iters=10)
PY
cut -c1-150 $OUT/nd2_radix32_ab.log

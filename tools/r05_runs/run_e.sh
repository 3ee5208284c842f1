#!/bin/bash
# round 5: the two-work-groups-per-transform form of the one-tile N-D shapes: parity, then A/B against the one-tile kernel
set -u
OUT=gpurun_out/r05e
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 python -m pytest tests/test_round5_gpu.py -x -q -k "two_work_groups" > $OUT/tests_nd2z.log 2>&1
echo "tests rc=$?" >> $OUT/tests_nd2z.log
tail -30 $OUT/tests_nd2z.log
python - > $OUT/nd2z_ab.log 2>&1 <<'PY'
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tools")
import numpy
from pyfft_amd import _native as N
import fused_sweep as fs
S32 = [(16, 16, 128), (32, 32, 32), (8, 64, 64), (128, 256), (256, 128), (32, 1024), (1024, 32), (512, 64)]
S64 = [(64, 16, 16), (16, 32, 32), (16, 16, 64), (128, 128), (64, 256), (256, 64), (32, 512), (512, 32)]
for gib in (0.03125, 0.25, 1.0):
    for alt in (6, 5):
        N.lib.mifft_debug_set(N.DEBUG_ALT_ROWS, alt)
        print("# %s GiB per side, %s" % (gib, "two work-groups per transform (fft_nd2z)" if alt == 5 else "one tile per CU (fft_nd2 huge)"), flush=True)
        for sh in S32:
            fs.sweep(sh, "complex64", gib, ["auto"], reps=5, iters=10)
        for sh in S64:
            fs.sweep(sh, "complex128", gib, ["auto"], reps=5, iters=10)
PY
cut -c1-150 $OUT/nd2z_ab.log

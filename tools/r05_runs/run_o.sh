#!/bin/bash
set -u
OUT=gpurun_out/r05o
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
python - > $OUT/nd2_radix16_f64.log 2>&1 <<'PY'
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tools")
import numpy
from pyfft_amd import _native as N
import fused_sweep as fs
S = [(16, 512), (16, 16, 32), (32, 16, 16), (16, 32, 16), (8, 16, 64), (16, 8, 64), (4, 16, 128), (16, 64, 8)]
print("# 1 GiB per side, fp64 two-per-CU tiles with 16-point y / z axes as ONE radix-16 stage (this build); compare profiles of the previous build", flush=True)
for sh in S:
    fs.sweep(sh, "complex128", 1.0, ["auto"], reps=5, iters=10)
PY
cut -c1-150 $OUT/nd2_radix16_f64.log

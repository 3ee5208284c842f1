#!/bin/bash
set -u
OUT=gpurun_out/r05u
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 python tools/fused_sweep.py 16x16x128 complex128 0.03125 auto@PYFFT_AMD_NO_OOP_ND=1,auto 16x16x128 complex128 0.25 auto@PYFFT_AMD_NO_OOP_ND=1,auto 16x16x128 complex128 1 auto@PYFFT_AMD_NO_OOP_ND=1,auto > $OUT/nd2z_f64_16_16_128.log 2>&1
cut -c1-150 $OUT/nd2z_f64_16_16_128.log

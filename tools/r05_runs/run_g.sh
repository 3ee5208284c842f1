#!/bin/bash
# round 5: four work-groups per transform (65536-point fp32 shapes) -- parity, then one launch against the plan's chain / persistent launch
set -u
OUT=gpurun_out/r05g
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 python -m pytest tests/test_round5_gpu.py -x -q -k "work_groups" > $OUT/tests_nd2z.log 2>&1
echo "tests rc=$?" >> $OUT/tests_nd2z.log
tail -30 $OUT/tests_nd2z.log
timeout 900 python tools/fused_sweep.py 256x256 complex64 0.03125 auto@PYFFT_AMD_NO_OOP_ND=1,auto 256x256 complex64 0.25 auto@PYFFT_AMD_NO_OOP_ND=1,auto 256x256 complex64 2 auto@PYFFT_AMD_NO_OOP_ND=1,auto \
  512x128 complex64 0.03125 auto@PYFFT_AMD_NO_OOP_ND=1,auto 512x128 complex64 2 auto@PYFFT_AMD_NO_OOP_ND=1,auto 128x512 complex64 0.03125 auto@PYFFT_AMD_NO_OOP_ND=1,auto 128x512 complex64 2 auto@PYFFT_AMD_NO_OOP_ND=1,auto \
  1024x64 complex64 2 auto@PYFFT_AMD_NO_OOP_ND=1,auto 64x1024 complex64 2 auto@PYFFT_AMD_NO_OOP_ND=1,auto 64x32x32 complex64 0.03125 auto@PYFFT_AMD_NO_OOP_ND=1,auto 64x32x32 complex64 2 auto@PYFFT_AMD_NO_OOP_ND=1,auto \
  32x32x64 complex64 0.03125 auto@PYFFT_AMD_NO_OOP_ND=1,auto 32x32x64 complex64 2 auto@PYFFT_AMD_NO_OOP_ND=1,auto 16x64x64 complex64 2 auto@PYFFT_AMD_NO_OOP_ND=1,auto 16x32x128 complex64 2 auto@PYFFT_AMD_NO_OOP_ND=1,auto > $OUT/nd2z_four_ab.log 2>&1
cut -c1-170 $OUT/nd2z_four_ab.log

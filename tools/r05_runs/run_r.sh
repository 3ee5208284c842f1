#!/bin/bash
# round 5: the 32768-point (fp32) / 16384-point (fp64) shapes WITHOUT a one-tile kernel: two launches (PYFFT_AMD_NO_OOP_ND=1) against
# one launch of two work-groups per transform, out of place
set -u
OUT=gpurun_out/r05r
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
V="auto@PYFFT_AMD_NO_OOP_ND=1,auto"
args=""
for g in 0.03125 1; do
  for s in 64x512 16x2048 2048x16 64x8x64 16x128x16 128x16x16 16x32x64 32x16x64 16x64x32 64x16x32 32x64x16 64x32x16; do args="$args $s complex64 $g $V"; done
  for s in 16x1024 1024x16 8x32x64 32x8x64 8x64x32 64x8x32 16x64x16 32x16x32 32x32x16 4x64x64 64x4x64; do args="$args $s complex128 $g $V"; done
done
timeout 1200 python tools/fused_sweep.py $args > $OUT/nd2z_more_shapes_ab.log 2>&1
cut -c1-150 $OUT/nd2z_more_shapes_ab.log

#!/bin/bash
# round 5: the default plan over a grid of 2-D and 3-D shapes beyond the one-launch tables, 1 GiB per side, fp32 and fp64: where are the outliers?
set -u
OUT=gpurun_out/r05grid
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
ARGS=""
for dt in complex64 complex128; do
  for ny in 128 256 512 1024 2048 4096; do for nx in 128 256 512 1024 2048 4096; do ARGS="$ARGS ${ny}x${nx} $dt 1 auto"; done; done
  for nz in 32 64 128 256; do for ny in 32 64 128 256; do for nx in 32 64 128 256; do
    if [ $((nz*ny*nx)) -ge 131072 ]; then ARGS="$ARGS ${nz}x${ny}x${nx} $dt 1 auto"; fi
  done; done; done
done
timeout 2400 python tools/fused_sweep.py $ARGS 2>&1 | cut -c1-150 > $OUT/grid.log
sort -k8 -n $OUT/grid.log | awk '{print $1, $2, $3, $4, $(NF-2)}' | sort -k5 -n | head -40

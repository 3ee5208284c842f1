#!/bin/bash
# round 5: store policy of the multi-launch plans in small launches (32 MiB): default (write-through) / non-temporal / plain, four dtypes
set -u
OUT=gpurun_out/r05st
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
V=auto,auto@MIFFT_STORE=1,auto@MIFFT_STORE=3
ARGS=""
for dt in complex64 complex128 float32 float64; do
  for shp in 1024x1024 128x128x128 32x32x128 16x16x128 65536 262144 1048576 512x512 256x256 64x64x64 2048x64 64x2048; do
    ARGS="$ARGS $shp $dt 0.03125 $V"
  done
done
timeout 1200 python tools/fused_sweep.py $ARGS 2>&1 | cut -c1-150 > $OUT/store_policy_small.log
python - <<'PY'
import re
rows=open('gpurun_out/r05st/store_policy_small.log').read().splitlines()
for i in range(0,len(rows)-2,3):
    f=[float(r.split()[-3]) if 'FAILED' not in r else 0 for r in rows[i:i+3]]
    flag = "  <-- not the best" if max(f[1:])>1.05*f[0] else ""
    print(rows[i][:40], " wt %.3f  nt %.3f  plain %.3f%s" % (f[0],f[1],f[2],flag))
PY

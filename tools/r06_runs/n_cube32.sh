#!/bin/bash
# round 6: (32, 32, 128) on the persistent two-pair kernel (y = 8 x 4): parity, then against the pipelined chunks at 1 GiB per side
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 python -m pytest tests/test_persistent_gpu.py -m gpu -q -x -k "fused_pair_small_axes" 2>&1 | tail -4
for dt in complex64 complex128; do
  b=$(( dt == complex64 ? 1024 : 512 )); [ $dt = complex128 ] && b=512 || b=1024
  for env in "" "PYFFT_AMD_NO_FUSEDP_ALT=1"; do
    echo "$dt x $b  $env"; env $env timeout 300 python3 tools/quick_bench.py one 32x32x128 $dt $b 2>&1 | grep "^(" | sed 's/passes=\[.*\]//' | cut -c1-150
  done
done

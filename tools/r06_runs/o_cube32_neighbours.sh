#!/bin/bash
# round 6: 3-D shapes with 32-point axes next to 64- / 128-point ones on the persistent two-pair kernel: parity, then against the pipelined
# chunks (PYFFT_AMD_NO_FUSEDP_ALT=1) at 1 GiB per side
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_persistent_gpu.py -m gpu -q -x -k "fused_pair_small_axes" 2>&1 | tail -4
for sh in 32x32x128 32x64x128 32x128x128 64x32x128 128x32x128 32x64x64 32x128x64; do
  n=$(python3 -c "import sys; a=[int(v) for v in '$sh'.split('x')]; print(a[0]*a[1]*a[2])")
  for dt in complex64 complex128; do
    [ $dt = complex64 ] && b=$(( (1<<30) / (n*8) )) || b=$(( (1<<30) / (n*16) ))
    new=$(timeout 300 python3 tools/quick_bench.py one $sh $dt $b 2>&1 | grep "^(" | grep -o "([0-9.]*% of" | tr -d '(% of')
    old=$(PYFFT_AMD_NO_FUSEDP_ALT=1 timeout 300 python3 tools/quick_bench.py one $sh $dt $b 2>&1 | grep "^(" | grep -o "([0-9.]*% of" | tr -d '(% of')
    echo "($sh) $dt x $b   persistent $new %   pipelined chunks $old %"
  done
done

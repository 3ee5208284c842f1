#!/bin/bash
# round 6: strided passes whose rows lie >= 2^16 points apart on 32-column tiles (default library); plane-fused route for transforms beyond
# half the cache only.  Parity of the affected tests, the probe again (A/B against 16-column tiles: MIFFT_NARROW_TILES=3), the longest 1-D sizes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06; mkdir -p $O
timeout 1200 python -m pytest tests/test_persistent_gpu.py tests/test_strided_gpu.py tests/test_pairs_gpu.py tests/test_full_size_gpu.py tests/test_random_sweep_gpu.py -m gpu -q -x --durations=8 > $O/j_tests.log 2>&1; tail -14 $O/j_tests.log
timeout 900 python3 tools/plane_fused_probe.py > $O/j_plane_fused_probe.log 2>&1; cat $O/j_plane_fused_probe.log
for sw in 0 3; do echo "MIFFT_NARROW_TILES=$sw"; MIFFT_NARROW_TILES=$sw timeout 600 python3 tools/quick_bench.py one 16777216 complex64 16 2>&1 | grep "^(" | cut -c1-40,110-200;  MIFFT_NARROW_TILES=$sw timeout 600 python3 tools/quick_bench.py one 8388608 complex64 32 2>&1 | grep "^(" | cut -c1-40,110-200; done > $O/j_long_1d.log 2>&1; cat $O/j_long_1d.log

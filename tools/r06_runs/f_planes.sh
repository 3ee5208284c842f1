#!/bin/bash
# round 6: dense split-complex N-D kernel (fft_nd2p): parity, then against the interleaved twins and the kernels it replaces
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_nd_gpu.py tests/test_strided_gpu.py -m gpu -q -x --durations=8 > $O/f_tests.log 2>&1; echo "rc=$?" >> $O/f_tests.log; tail -15 $O/f_tests.log
timeout 900 python -m pytest tests/test_errors_gpu.py tests/test_functionality_gpu.py tests/test_random_sweep_gpu.py tests/test_generic_gpu.py -m gpu -q -x -k "float32 or float64 or f32 or f64 or split or random or tiled" > $O/f_tests2.log 2>&1; echo "rc=$?" >> $O/f_tests2.log; tail -4 $O/f_tests2.log
timeout 1200 python3 tools/planes_probe.py > $O/f_planes_probe.log 2>&1; cat $O/f_planes_probe.log

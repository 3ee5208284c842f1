#!/bin/bash
# round 6: float32 planes of the 32768-point shapes on two half-size work-groups per transform (csrc/fft_nd2zp.hpp): parity, then against
# the routes before (MIFFT_DEBUG_ALT_ROWS = 7: one 256 KiB tile per CU / two launches) and the interleaved twins at 1 GiB and 32 MiB
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06; mkdir -p $O
timeout 1200 python -m pytest tests/test_nd_gpu.py tests/test_errors_gpu.py -m gpu -q -x --durations=5 > $O/k_tests.log 2>&1; tail -12 $O/k_tests.log
timeout 900 python3 tools/planes_probe.py halves > $O/k_planes_halves.log 2>&1; cat $O/k_planes_halves.log

#!/bin/bash
# round 4, run ao: the final split defaults (tests), and interleaved 2^20 on the lane-interleaved double tiles (experiment)
mkdir -p gpurun_out/r04an
timeout 900 python -m pytest tests/test_round4_gpu.py tests/test_round2_gpu.py tests/test_errors_gpu.py -q -x -k "split or strateg or fused" 2>&1 | tail -5 > gpurun_out/r04an/tests2.log
timeout 900 python tools/fused_sweep.py 1048576 complex64 8 auto,auto@MIFFT_NARROW_TILES=3,f:7:14@MIFFT_NARROW_TILES=3,auto,auto@MIFFT_NARROW_TILES=3 1048576 complex64 2 auto,auto@MIFFT_NARROW_TILES=3 \
   1048576 float32 2 auto,auto@MIFFT_STORE=3 262144 float32 2 auto,f:56:112 65536 float32 2 auto > gpurun_out/r04an/sweep2.log 2>&1
cat gpurun_out/r04an/tests2.log; cat gpurun_out/r04an/sweep2.log

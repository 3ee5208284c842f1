#!/bin/bash
# round 4, run bg: interleaved (512, 2048) and (1024, 2048) on the row-first persistent kernel
mkdir -p gpurun_out/r04bg
timeout 900 python -m pytest tests/test_round4_gpu.py -q -x -k "rectangles" 2>&1 | tail -4 > gpurun_out/r04bg/tests.log
timeout 900 python tools/fused_sweep.py 512x2048 complex64 2 auto,f:8:14,f:16:28,auto@MIFFT_NARROW_TILES=4,pipelined 1024x2048 complex64 2 auto,f:8:14,auto@MIFFT_NARROW_TILES=4,pipelined \
   512x2048 complex64 8 auto,pipelined 1024x2048 complex64 8 auto,auto@MIFFT_NARROW_TILES=4 > gpurun_out/r04bg/sweep.log 2>&1
cat gpurun_out/r04bg/tests.log; cat gpurun_out/r04bg/sweep.log

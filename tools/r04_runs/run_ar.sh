#!/bin/bash
# round 4, run ar: split-complex fp32 plain launches on the lane-interleaved double tiles (chain mode and pipelined chunks)
mkdir -p gpurun_out/r04aq
timeout 900 python -m pytest tests/test_errors_gpu.py tests/test_round2_gpu.py tests/test_round4_gpu.py tests/test_functionality_gpu.py -q -x -k "split or float32 or f32 or strateg" 2>&1 | tail -5 > gpurun_out/r04aq/tests.log
timeout 900 python tools/fused_sweep.py 1048576 float32 0.125 auto,auto@MIFFT_NARROW_TILES=1 262144 float32 0.125 auto,auto@MIFFT_NARROW_TILES=1 65536 float32 0.125 auto,auto@MIFFT_NARROW_TILES=1 \
  1024x1024 float32 0.125 auto,auto@MIFFT_NARROW_TILES=1 1048576 float32 0.03125 auto,auto@MIFFT_NARROW_TILES=1 1024x1024 float32 0.03125 auto,auto@MIFFT_NARROW_TILES=1 \
  512x512 float32 0.125 auto,auto@MIFFT_NARROW_TILES=1 2048x2048 float32 2 auto,auto@MIFFT_NARROW_TILES=1 256x256 float32 2 auto,auto@MIFFT_NARROW_TILES=1 > gpurun_out/r04aq/sweep2.log 2>&1
cat gpurun_out/r04aq/tests.log; cat gpurun_out/r04aq/sweep2.log

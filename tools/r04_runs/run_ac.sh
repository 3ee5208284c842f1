#!/bin/bash
# round 4, run ac: split-complex fp32 with the sibling tiles side by side in one 512-thread work-group
mkdir -p gpurun_out/r04ac
timeout 900 python -m pytest tests/test_round4_gpu.py tests/test_round2_gpu.py tests/test_errors_gpu.py -q -x -k "split or strateg or fused" 2>&1 | tail -8 > gpurun_out/r04ac/tests.log
timeout 900 python tools/fused_sweep.py 1048576 float32 2 auto,f:14:28,f:14:28@MIFFT_NARROW_TILES=1,pipelined 524288 float32 2 auto,f:28:56@MIFFT_NARROW_TILES=1 \
   262144 float32 2 auto,f:56:112,f:28:56 131072 float32 2 auto,f:112:224,f:56:112 65536 float32 2 auto,f:112:224 \
   1024x1024 float32 2 auto,f:14:28,f:14:28@MIFFT_NARROW_TILES=1 512x512 float32 2 auto,f:56:112,f:56:112@MIFFT_NARROW_TILES=1 > gpurun_out/r04ac/sweep.log 2>&1
cat gpurun_out/r04ac/tests.log; tail -40 gpurun_out/r04ac/sweep.log

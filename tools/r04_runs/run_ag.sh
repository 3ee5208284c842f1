#!/bin/bash
# round 4, run ag: split-complex fp32 2-D on the row-first persistent kernel
mkdir -p gpurun_out/r04ag
timeout 900 python -m pytest tests/test_round4_gpu.py -q -x -k "row_first" 2>&1 | tail -8 > gpurun_out/r04ag/tests.log
timeout 900 python tools/fused_sweep.py 1024x1024 float32 2 auto,auto@PYFFT_AMD_NO_SPLIT_ROWFIRST=1,f:7:14,f:28:56 512x512 float32 2 auto,auto@PYFFT_AMD_NO_SPLIT_ROWFIRST=1,f:56:112 \
   256x256 float32 2 auto,auto@PYFFT_AMD_NO_SPLIT_ROWFIRST=1,f:112:224 512x1024 float32 2 auto,auto@PYFFT_AMD_NO_SPLIT_ROWFIRST=1 1024x512 float32 2 auto,auto@PYFFT_AMD_NO_SPLIT_ROWFIRST=1 \
   256x1024 float32 2 auto,auto@PYFFT_AMD_NO_SPLIT_ROWFIRST=1 1024x256 float32 2 auto,auto@PYFFT_AMD_NO_SPLIT_ROWFIRST=1 1024x1024 float32 0.5 auto,auto@PYFFT_AMD_NO_SPLIT_ROWFIRST=1 > gpurun_out/r04ag/sweep.log 2>&1
cat gpurun_out/r04ag/tests.log; tail -40 gpurun_out/r04ag/sweep.log

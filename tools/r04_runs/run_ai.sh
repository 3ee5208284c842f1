#!/bin/bash
mkdir -p gpurun_out/r04ag
timeout 900 python -m pytest tests/test_round4_gpu.py tests/test_round2_gpu.py tests/test_errors_gpu.py tests/test_functionality_gpu.py -q -x -k "split or row_first or strateg or float32 or f32" 2>&1 | tail -5 > gpurun_out/r04ag/tests2.log
timeout 600 python tools/fused_sweep.py 1024x1024 float32 2 auto,auto@PYFFT_AMD_NO_SPLIT_ROWFIRST=1 512x512 float32 2 auto 1024x512 float32 2 auto 512x1024 float32 2 auto 1024x1024 float32 0.5 auto,auto@PYFFT_AMD_NO_SPLIT_ROWFIRST=1 1024x1024 float32 4 auto,auto@PYFFT_AMD_NO_SPLIT_ROWFIRST=1 > gpurun_out/r04ag/sweep3.log 2>&1
cat gpurun_out/r04ag/tests2.log; cat gpurun_out/r04ag/sweep3.log

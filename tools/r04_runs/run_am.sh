#!/bin/bash
mkdir -p gpurun_out/r04aj
timeout 900 python -m pytest tests/test_round4_gpu.py tests/test_round2_gpu.py tests/test_round3_gpu.py tests/test_errors_gpu.py -q -x -k "split or float64 or f64 or strateg or fp64" 2>&1 | tail -5 > gpurun_out/r04aj/tests4.log
timeout 900 python tools/fused_sweep.py 65536 float64 2 auto,pipelined 131072 float64 2 auto,pipelined 262144 float64 2 auto,pipelined 524288 float64 2 auto,pipelined 1048576 float64 2 auto,pipelined \
  1024x1024 float64 2 auto,pipelined 2048x2048 float32 2 auto,f:4:7 > gpurun_out/r04aj/sweep4.log 2>&1
cat gpurun_out/r04aj/tests4.log; cat gpurun_out/r04aj/sweep4.log

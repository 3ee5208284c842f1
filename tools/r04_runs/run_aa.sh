#!/bin/bash
# round 4, run aa: split-complex planes on the persistent two-pair kernel
mkdir -p gpurun_out/r04aa
timeout 900 python -m pytest tests/test_round4_gpu.py -q -x -k "fused_pair" 2>&1 | tail -8 > gpurun_out/r04aa/tests.log
timeout 900 python tools/fused_sweep.py 64x64x64 float32 2 auto,pipelined 64x128x128 float32 2 auto,pipelined 64x64x128 float32 2 auto,pipelined \
   128x128x128 float32 2 auto 64x64x64 float64 2 auto,pipelined 128x128x128 float64 2 auto,pipelined 64x128x128 float64 2 auto,pipelined \
   128x128x64 float64 2 auto,pipelined 128x64x64 float64 2 auto,pipelined > gpurun_out/r04aa/sweep.log 2>&1
cat gpurun_out/r04aa/tests.log; tail -40 gpurun_out/r04aa/sweep.log

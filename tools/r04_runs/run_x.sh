#!/bin/bash
# round 4, run x: split-complex fp32 with sibling tiles per item; 2-D shapes with a 256-point axis on the persistent kernels
mkdir -p gpurun_out/r04x
python -m pytest tests/test_round4_gpu.py -q -x -k "256_sides or sibling or per_xcd" 2>&1 | tail -8 > gpurun_out/r04x/tests.log
python tools/fused_sweep.py 1048576 float32 2 auto,f:14:28,pipelined 524288 float32 2 auto 262144 float32 2 auto,x:3:6 65536 float32 2 auto,f:112:224,x:8:16 \
   131072 float32 2 auto,f:112:224 256x256 complex64 2 auto,pipelined 256x512 complex64 2 auto,pipelined 512x256 complex64 2 auto,pipelined \
   256x1024 complex64 2 auto,pipelined 1024x256 complex64 2 auto,pipelined 256x256 complex128 2 auto,pipelined 256x512 complex128 2 auto,pipelined \
   512x256 complex128 2 auto,pipelined > gpurun_out/r04x/sweep.log 2>&1
cat gpurun_out/r04x/tests.log; tail -40 gpurun_out/r04x/sweep.log

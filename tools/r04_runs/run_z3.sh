#!/bin/bash
# round 4, run z3: all {64, 128}^3 shapes on the persistent two-pair kernel (tests + defaults against the pipelined chunks)
mkdir -p gpurun_out/r04z
timeout 900 python -m pytest tests/test_round4_gpu.py -q -x -k "small_axes or cube_128" 2>&1 | tail -8 > gpurun_out/r04z/tests3.log
timeout 900 python tools/fused_sweep.py 64x128x64 complex64 2 auto,pipelined 64x128x64 complex128 2 auto,pipelined 128x64x128 complex64 2 auto,pipelined \
   128x64x128 complex128 2 auto,pipelined 64x64x128 complex64 2 auto,pipelined 64x64x128 complex128 2 auto,pipelined 128x64x64 complex64 2 auto,pipelined \
   128x64x64 complex128 2 auto,pipelined 64x64x64 complex64 2 auto 64x64x64 complex128 2 auto 64x128x128 complex64 2 auto 128x128x64 complex64 2 auto \
   128x128x128 complex64 2 auto 128x128x128 complex128 2 auto > gpurun_out/r04z/sweep3.log 2>&1
cat gpurun_out/r04z/tests3.log; tail -40 gpurun_out/r04z/sweep3.log

#!/bin/bash
# round 4, run z: the persistent two-pair kernel on 3-D shapes with 64- and 128-point axes
mkdir -p gpurun_out/r04z
timeout 600 python -m pytest tests/test_round4_gpu.py -q -x -k "small_axes or generic_plans or split_planes_on" 2>&1 | tail -8 > gpurun_out/r04z/tests.log
timeout 900 python tools/fused_sweep.py 64x64x64 complex64 2 auto,pipelined,f:4:7,f:16:28,f:32:56 64x64x64 complex128 2 auto,pipelined,f:16:28 \
   64x128x128 complex64 2 auto,pipelined,f:4:7 64x128x128 complex128 2 auto,pipelined 128x128x64 complex64 2 auto,pipelined,f:4:7 \
   128x128x64 complex128 2 auto,pipelined 64x64x64 complex64 0.5 auto,pipelined 64x128x128 complex64 0.5 auto,pipelined > gpurun_out/r04z/sweep.log 2>&1
cat gpurun_out/r04z/tests.log; tail -40 gpurun_out/r04z/sweep.log

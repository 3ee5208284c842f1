#!/bin/bash
# round 4, run z2: ring sizes of the persistent two-pair kernel on the small 3-D shapes
mkdir -p gpurun_out/r04z
timeout 900 python tools/fused_sweep.py 64x64x64 complex64 2 f:32:56,f:64:112,f:48:112,f:80:112 64x64x64 complex128 2 f:16:28,f:32:56,f:24:56 \
   64x128x128 complex64 2 f:8:14,f:16:28,f:12:28,f:20:28 64x128x128 complex128 2 f:8:14,f:6:14,f:10:14 128x128x64 complex64 2 f:16:28,f:12:28 \
   128x128x64 complex128 2 f:8:14,f:10:14 64x64x64 complex64 0.5 f:32:56,f:64:112 64x128x128 complex64 0.5 f:16:28 > gpurun_out/r04z/sweep2.log 2>&1
tail -40 gpurun_out/r04z/sweep2.log

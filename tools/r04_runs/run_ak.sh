#!/bin/bash
mkdir -p gpurun_out/r04aj
timeout 900 python tools/fused_sweep.py 1024x1024 float32 2 f:7:14,f:14:28 512x512 float32 2 f:28:56,f:56:112 1024x512 float32 2 f:14:28,f:28:56 512x1024 float32 2 f:14:28,f:28:56 \
  1024x256 float32 2 f:28:56,f:56:112 256x1024 float32 2 f:28:56,f:56:112,auto 256x512 float32 2 f:56:112,f:112:224 512x256 float32 2 f:56:112,f:112:224 256x256 float32 2 f:112:224,auto \
  1024x1024 float32 0.5 f:7:14,f:14:28,pipelined 512x512 float32 0.5 f:28:56,f:56:112,pipelined > gpurun_out/r04aj/sweep2.log 2>&1
tail -40 gpurun_out/r04aj/sweep2.log

#!/bin/bash
mkdir -p gpurun_out/r04ag
timeout 900 python tools/fused_sweep.py 1024x1024 float32 2 f:4:8,f:7:14,f:14:28,auto@PYFFT_AMD_NO_SPLIT_ROWFIRST=1 512x512 float32 2 f:14:28,f:28:56,f:56:112 \
   256x256 float32 2 f:28:56,f:56:112,f:112:224,f:200:400 512x1024 float32 2 f:14:28,f:28:56 1024x512 float32 2 f:7:14,f:14:28,f:28:56 \
   256x1024 float32 2 f:14:28,f:28:56,f:56:112 1024x256 float32 2 f:14:28,f:28:56,f:56:112 256x512 float32 2 f:28:56,f:56:112,auto@PYFFT_AMD_NO_SPLIT_ROWFIRST=1 512x256 float32 2 f:28:56,f:56:112,auto@PYFFT_AMD_NO_SPLIT_ROWFIRST=1 > gpurun_out/r04ag/sweep2.log 2>&1
tail -40 gpurun_out/r04ag/sweep2.log

#!/bin/bash
mkdir -p gpurun_out/r04an
timeout 900 python tools/fused_sweep.py 131072 float32 2 auto,f:56:112 1024x1024 float32 2 auto,f:14:28@PYFFT_AMD_NO_SPLIT_ROWFIRST=1 512x512 float32 2 auto,f:56:112@PYFFT_AMD_NO_SPLIT_ROWFIRST=1 \
  65536 float32 0.5 auto 1048576 float32 0.5 auto,pipelined 1048576 float32 8 auto > gpurun_out/r04an/sweep3.log 2>&1
cat gpurun_out/r04an/sweep3.log

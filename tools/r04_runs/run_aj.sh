#!/bin/bash
# round 4, run aj: split-complex fp32 siblings with non-temporal accesses to the planes, ring sizes; the row-first kernel with non-temporal stores
mkdir -p gpurun_out/r04aj
timeout 900 python tools/fused_sweep.py 1048576 float32 2 auto,auto@MIFFT_STORE=1,f:7:14,f:7:14@MIFFT_STORE=1,f:10:20@MIFFT_STORE=1 524288 float32 2 auto,auto@MIFFT_STORE=1,f:14:28,f:14:28@MIFFT_STORE=1 \
  65536 float32 2 auto,auto@MIFFT_STORE=1 131072 float32 2 auto,auto@MIFFT_STORE=1,f:112:224@MIFFT_STORE=1 262144 float32 2 f:28:56,f:28:56@MIFFT_STORE=1,f:56:112@MIFFT_STORE=1 \
  1024x1024 float32 2 auto,f:14:28 512x512 float32 2 auto,f:56:112 1024x512 float32 2 auto 256x256 float32 2 f:56:112,f:112:224 > gpurun_out/r04aj/sweep.log 2>&1
tail -40 gpurun_out/r04aj/sweep.log

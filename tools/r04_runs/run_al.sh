#!/bin/bash
# round 4, run al: non-temporal accesses to the planes in the 512-thread tiles (fp64 split 2^20 and 1024^2, fp32 split 2^22)
mkdir -p gpurun_out/r04aj
timeout 900 python tools/fused_sweep.py 1048576 float64 2 auto,auto@MIFFT_STORE=1,f:4:7,f:4:7@MIFFT_STORE=1,pipelined 1024x1024 float64 2 auto,f:8:14,f:8:14@MIFFT_STORE=1,f:4:7@MIFFT_STORE=1 \
   4194304 float32 2 auto,auto@MIFFT_STORE=1,pipelined > gpurun_out/r04aj/sweep3.log 2>&1
tail -20 gpurun_out/r04aj/sweep3.log

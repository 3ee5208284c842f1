#!/bin/bash
mkdir -p gpurun_out/r04aa
timeout 900 python tools/fused_sweep.py 128x128x128 float32 2 auto,pipelined 128x128x64 float32 2 auto,pipelined 128x64x128 float32 2 auto,pipelined 128x64x64 float32 2 auto,pipelined > gpurun_out/r04aa/sweep2.log 2>&1
tail -10 gpurun_out/r04aa/sweep2.log

#!/bin/bash
# round 4, run u: split-complex layouts on the two-pass shapes -- persistent kernel against the pipelined chunks
mkdir -p gpurun_out/r04u
python tools/fused_sweep.py 1048576 float32 2 auto,pipelined,f:14:28 262144 float32 2 auto,pipelined 65536 float32 2 auto,pipelined \
   1024x1024 float32 2 auto,pipelined,f:14:28 1048576 float64 2 auto,pipelined 128x128x128 float32 2 auto 512x512 float32 2 auto,f:56:112 \
   > gpurun_out/r04u/split_sweep.log 2>&1
tail -30 gpurun_out/r04u/split_sweep.log

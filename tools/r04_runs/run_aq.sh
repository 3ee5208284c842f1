#!/bin/bash
# round 4, run aq: split-complex against interleaved below the chain threshold (plain launches over the whole batch)
mkdir -p gpurun_out/r04aq
timeout 900 python tools/fused_sweep.py 1048576 complex64 0.125 auto 1048576 float32 0.125 auto 262144 complex64 0.125 auto 262144 float32 0.125 auto 65536 complex64 0.125 auto 65536 float32 0.125 auto \
  1024x1024 complex64 0.125 auto 1024x1024 float32 0.125 auto 1048576 complex64 0.03125 auto 1048576 float32 0.03125 auto 1024x1024 complex64 0.03125 auto 1024x1024 float32 0.03125 auto \
  128x128x128 complex64 0.125 auto 128x128x128 float32 0.125 auto 1048576 complex128 0.125 auto 1048576 float64 0.125 auto > gpurun_out/r04aq/sweep.log 2>&1
cat gpurun_out/r04aq/sweep.log

#!/bin/bash
# round 4, run at: single-pass rows, split-complex against interleaved
mkdir -p gpurun_out/r04at
timeout 900 python tools/fused_sweep.py 1024 complex64 1 auto 1024 float32 1 auto 4096 complex64 1 auto 4096 float32 1 auto 8192 complex64 1 auto 8192 float32 1 auto 32768 complex64 1 auto 32768 float32 1 auto \
  256 complex64 1 auto 256 float32 1 auto 1024 complex128 1 auto 1024 float64 1 auto 4096 complex128 1 auto 4096 float64 1 auto 1024 complex64 0.03125 auto 1024 float32 0.03125 auto \
  8192 complex64 0.03125 auto 8192 float32 0.03125 auto 128x128 complex64 1 auto 128x128 float32 1 auto 16x16x16 complex64 1 auto 16x16x16 float32 1 auto > gpurun_out/r04at/rows.log 2>&1
cat gpurun_out/r04at/rows.log

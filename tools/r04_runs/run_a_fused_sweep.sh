#!/bin/bash
# round 4, call a: lag / ring / work-list sweep of the persistent two-pass kernels at 1 GiB and 8 GiB per side, after the
# non-temporal streaming form became the default for every interleaved size.  gpurun --timeout 1500 -- 'bash tools/r04_runs/run_a_fused_sweep.sh'
set -u
OUT=gpurun_out/r04a
mkdir -p $OUT
S="python3 tools/fused_sweep.py"
for G in 1 8; do
timeout 1200 $S \
  65536    complex64 $G auto,f:56:112,f:112:224,f:224:448,x:4:8:0,x:8:16:0,x:16:32:0 \
  131072   complex64 $G auto,f:28:56,f:56:112,f:112:224,x:4:8:0,x:8:16:0 \
  262144   complex64 $G auto,pipelined,f:14:28,f:21:42,f:28:56,f:42:84,f:56:112,f:40:56,x:3:6:0,x:4:8:0,x:6:12:0,x:4:8:2 \
  524288   complex64 $G auto,pipelined,f:7:14,f:10:20,f:14:28,f:21:42,f:28:56,f:20:28,f:10:28,x:3:6:0,x:4:8:0,x:4:7:0,x:6:12:0,x:28:56:2 \
  1048576  complex64 $G auto,f:10:20,f:12:24,f:14:28,f:18:28,f:10:28,f:16:32,x:2:4:0,x:3:4:0 \
  2097152  complex64 $G auto,pipelined,f:4:8,f:6:12,f:7:14,f:8:14,f:5:14,f:10:14 \
  4194304  complex64 $G auto,pipelined,f:3:6,f:3:7,f:4:7,f:5:7,f:4:8 \
  512x512  complex64 $G auto,pipelined,f:14:28,f:28:56,f:56:112 \
  1024x1024 complex64 $G auto,f:10:20,f:14:28,f:18:28 \
  2048x2048 complex64 $G auto,f:3:6,f:4:7,f:5:7 \
  1048576  complex128 $G auto,pipelined,f:6:12,f:8:14,f:10:14,f:5:14 \
  1024x1024 complex128 $G auto,f:6:12,f:8:14,f:10:14 \
  >> $OUT/fused_sweep.log 2>&1
done
tail -5 $OUT/fused_sweep.log

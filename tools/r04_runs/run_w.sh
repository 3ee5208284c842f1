#!/bin/bash
# round 4, run w: split-complex 1-D sizes on the per-XCD work lists (sibling tiles share 128-byte lines within one L2)
mkdir -p gpurun_out/r04w
python tools/fused_sweep.py 1048576 float32 2 auto,x:1:3,x:2:3,x:1:2 262144 float32 2 auto,x:4:8,x:6:12,x:3:6 65536 float32 2 auto,x:8:16,x:16:32 \
   524288 float32 2 auto,x:3:6,x:2:4 131072 float32 2 auto,x:8:16 > gpurun_out/r04w/split_xcd.log 2>&1
python -m pytest tests/test_round4_gpu.py -q -x -k "per_xcd" 2>&1 | tail -3 >> gpurun_out/r04w/split_xcd.log
tail -30 gpurun_out/r04w/split_xcd.log

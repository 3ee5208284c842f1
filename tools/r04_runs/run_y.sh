#!/bin/bash
# round 4, run y: split planes on the per-XCD lists by default; the 256-side 2-D shapes; round-4 test file
mkdir -p gpurun_out/r04y
python -m pytest tests/test_round4_gpu.py tests/test_host.py -q -x -m gpu 2>&1 | tail -5 > gpurun_out/r04y/tests.log
python tools/fused_sweep.py 65536 float32 2 auto 131072 float32 2 auto 262144 float32 2 auto 65536 float32 0.5 auto 262144 float32 0.5 auto > gpurun_out/r04y/sweep.log 2>&1
cat gpurun_out/r04y/tests.log gpurun_out/r04y/sweep.log

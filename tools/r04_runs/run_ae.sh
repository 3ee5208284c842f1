#!/bin/bash
mkdir -p gpurun_out/r04ac
timeout 900 python -m pytest tests/test_round4_gpu.py -q -x -k "split" 2>&1 | tail -5 > gpurun_out/r04ac/tests3.log
timeout 900 python tools/fused_sweep.py 65536 float32 2 auto,f:112:224,f:28:56,auto 131072 float32 2 auto,f:112:224,f:28:56,auto 65536 float32 0.5 auto,pipelined 131072 float32 0.5 auto,pipelined \
  2097152 float32 2 auto,pipelined > gpurun_out/r04ac/sweep3.log 2>&1
cat gpurun_out/r04ac/tests3.log; tail -20 gpurun_out/r04ac/sweep3.log

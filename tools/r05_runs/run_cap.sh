#!/bin/bash
# round 5: the capture tests over and over (the pipelined chunks are recorded as a linear graph now), then the soak suite again
set -u
OUT=gpurun_out/r05cap
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  timeout 600 python -m pytest tests/test_round5_gpu.py -m gpu -x -q -k "captured_execute or torch_cuda_graph or four_host_threads" 2>&1 | tail -1
done > $OUT/capture_loop.log 2>&1
cat $OUT/capture_loop.log
PYFFT_AMD_SWEEP=1 timeout 3000 python -m pytest tests -m gpu -q --durations=8 > $OUT/tests_gpu_sweep.log 2>&1
echo "tests rc=$?" >> $OUT/tests_gpu_sweep.log
tail -12 $OUT/tests_gpu_sweep.log

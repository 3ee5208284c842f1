#!/bin/bash
# round 4, call d: GPU suite again (test fixes, N-D smooth kernel, big Bluestein rows), then the remaining sweeps.
set -u
OUT=gpurun_out/r04d
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python3 -m pytest tests/test_round4_gpu.py -q -m gpu > $OUT/pytest_r4.log 2>&1; tail -25 $OUT/pytest_r4.log
timeout 1500 python3 -m pytest tests -x -q -m gpu --deselect tests/test_round4_gpu.py > $OUT/pytest_all.log 2>&1; tail -5 $OUT/pytest_all.log
S="python3 tools/fused_sweep.py"
timeout 900 $S \
  131072  complex64 0.5 auto,auto@PYFFT_AMD_NO_FUSEDX=1 \
  131072  complex64 2 auto,auto@PYFFT_AMD_NO_FUSEDX=1,x:4:8,x:16:32 \
  131072  complex64 8 auto,auto@PYFFT_AMD_NO_FUSEDX=1 \
  65536   complex64 8 auto,auto@PYFFT_AMD_NO_FUSEDX=1 \
  262144  complex64 0.5 auto,x:4:8 \
  262144  complex64 2 auto,x:4:8 \
  524288  complex64 0.5 auto,x:4:8 \
  524288  complex64 2 auto,x:4:8 \
  > $OUT/list_sweep.log 2>&1
cat $OUT/list_sweep.log
# fp64 long 1-D: what the pipelined chunks can do (streams, chunk size) before any new kernel
timeout 900 $S \
  2097152 complex128 4 auto,auto@PYFFT_AMD_PIPE_STREAMS=1,auto@PYFFT_AMD_PIPE_STREAMS=3,auto@PYFFT_AMD_PIPE_MB=32,auto@PYFFT_AMD_PIPE_MB=128,chain \
  4194304 complex128 4 auto,auto@PYFFT_AMD_PIPE_STREAMS=1,auto@PYFFT_AMD_PIPE_STREAMS=3,auto@PYFFT_AMD_PIPE_MB=128,chain \
  > $OUT/fp64_long.log 2>&1
cat $OUT/fp64_long.log
timeout 600 python3 tools/mixed_probe.py > $OUT/mixed.log 2>&1; tail -40 $OUT/mixed.log

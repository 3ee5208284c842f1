#!/bin/bash
# round 4, call c: GPU suite on the tree with the 512-thread two-pair kernel, write-through stealing lists, the statically dealt
# sequential list and the rectangular 2-D kernels; then the sweeps.  gpurun --timeout 2700 -- 'bash tools/r04_runs/run_c.sh'
set -u
OUT=gpurun_out/r04c
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python3 -m pytest tests/test_round4_gpu.py -q -m gpu > $OUT/pytest_r4.log 2>&1; tail -25 $OUT/pytest_r4.log
timeout 1500 python3 -m pytest tests -x -q -m gpu --deselect tests/test_round4_gpu.py > $OUT/pytest_all.log 2>&1; tail -5 $OUT/pytest_all.log
S="python3 tools/fused_sweep.py"
timeout 600 $S \
  128x128x128 complex64 1 pipelined,auto,f:4:8,f:6:12 \
  128x128x128 complex64 4 pipelined,auto,f:4:8,f:6:12,f:4:14 \
  128x128x128 complex128 1 pipelined,auto \
  128x128x128 complex128 4 pipelined,auto \
  > $OUT/cube_sweep.log 2>&1
cat $OUT/cube_sweep.log
timeout 900 $S \
  65536   complex64 0.5 auto,auto@PYFFT_AMD_NO_FUSEDX=1 \
  65536   complex64 2 auto,auto@PYFFT_AMD_NO_FUSEDX=1,x:4:8 \
  65536   complex64 8 auto,auto@PYFFT_AMD_NO_FUSEDX=1 \
  131072  complex64 0.5 auto,auto@PYFFT_AMD_NO_FUSEDX=1 \
  131072  complex64 2 auto,auto@PYFFT_AMD_NO_FUSEDX=1,x:4:8 \
  131072  complex64 8 auto,auto@PYFFT_AMD_NO_FUSEDX=1 \
  262144  complex64 0.5 auto,x:4:8 \
  262144  complex64 2 auto,x:4:8 \
  524288  complex64 0.5 auto,x:4:8 \
  524288  complex64 2 auto,x:4:8 \
  > $OUT/list_sweep.log 2>&1
cat $OUT/list_sweep.log
timeout 900 $S \
  512x1024 complex64 1 pipelined,auto \
  512x1024 complex64 4 pipelined,auto \
  1024x512 complex64 1 pipelined,auto \
  1024x512 complex64 4 pipelined,auto \
  1024x2048 complex64 1 pipelined,auto \
  1024x2048 complex64 4 pipelined,auto,auto@PYFFT_AMD_FUSED_WGS=2 \
  2048x1024 complex64 1 pipelined,auto \
  2048x1024 complex64 4 pipelined,auto \
  512x2048 complex64 4 pipelined,auto \
  2048x512 complex64 4 pipelined,auto \
  > $OUT/rect_sweep.log 2>&1
cat $OUT/rect_sweep.log
timeout 600 $S \
  1024x1024 complex64 0.03125 auto,seq \
  1024x1024 complex128 0.03125 auto,seq \
  128x128x128 complex64 0.03125 auto,seq \
  128x128x128 complex128 0.03125 auto,seq \
  1048576 complex64 0.03125 auto,seq \
  262144 complex64 0.03125 auto,seq \
  4194304 complex64 0.03125 auto,seq \
  1024x1024 complex64 0.0625 auto,seq \
  1024x1024 complex64 0.125 auto,seq \
  128x128x128 complex64 0.125 auto,seq \
  1048576 complex64 0.125 auto,seq \
  1024x1024 complex64 0.25 auto,seq \
  > $OUT/small_sweep.log 2>&1
cat $OUT/small_sweep.log

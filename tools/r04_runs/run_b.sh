#!/bin/bash
# round 4, call b: the whole GPU suite on the new tree (persistent two-pair kernel, per-XCD lists with work stealing, alternating
# counter sets, sequential lists), then the sweeps that decide the defaults.  gpurun --timeout 2400 -- 'bash tools/r04_runs/run_b.sh'
set -u
OUT=gpurun_out/r04b
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 python3 -m pytest tests/test_round4_gpu.py -x -q -m gpu > $OUT/pytest_r4.log 2>&1; tail -15 $OUT/pytest_r4.log
timeout 1200 python3 -m pytest tests -x -q -m gpu --deselect tests/test_round4_gpu.py > $OUT/pytest_all.log 2>&1; tail -5 $OUT/pytest_all.log
S="python3 tools/fused_sweep.py"
# (1) the cubes: chain / pipelined / persistent two-pair launch by ring
timeout 900 $S \
  128x128x128 complex64 1 pipelined,auto,f:2:4,f:3:6,f:4:8,f:6:12,f:7:14,f:4:14 \
  128x128x128 complex64 4 pipelined,auto,f:2:4,f:3:6,f:4:8,f:6:12,f:7:14,f:4:14 \
  128x128x128 complex128 1 pipelined,auto,f:2:4,f:3:6,f:4:7,f:3:7,f:5:7 \
  128x128x128 complex128 4 pipelined,auto,f:2:4,f:3:6,f:4:7,f:3:7,f:5:7 \
  > $OUT/cube_sweep.log 2>&1
cat $OUT/cube_sweep.log
# (2) per-XCD lists after work stealing, the ring rule, memset A/B, by buffer size
timeout 1200 $S \
  65536   complex64 0.5 auto,auto@PYFFT_AMD_NO_FUSEDX=1 \
  65536   complex64 2 auto,auto@PYFFT_AMD_NO_FUSEDX=1,x:16:32:0 \
  131072  complex64 0.5 auto,auto@PYFFT_AMD_NO_FUSEDX=1 \
  131072  complex64 2 auto,auto@PYFFT_AMD_NO_FUSEDX=1,x:16:32:0 \
  131072  complex64 8 auto,auto@PYFFT_AMD_NO_FUSEDX=1 \
  262144  complex64 0.5 auto,pipelined,x:4:8:0 \
  262144  complex64 2 auto,pipelined,x:4:8:0 \
  524288  complex64 0.5 auto,pipelined,x:4:8:0 \
  524288  complex64 1 auto,auto@PYFFT_AMD_FUSED_MEMSET=1,x:4:8:0,f:14:28 \
  524288  complex64 2 auto,pipelined,x:4:8:0 \
  524288  complex64 8 auto,x:4:8:0 \
  1048576 complex64 0.5 auto,pipelined \
  1048576 complex64 1 auto,auto@PYFFT_AMD_FUSED_MEMSET=1 \
  1048576 complex64 2 auto \
  2097152 complex64 1 auto,auto@PYFFT_AMD_FUSED_WGS=1 \
  2097152 complex64 8 auto,auto@PYFFT_AMD_FUSED_WGS=1 \
  > $OUT/list_sweep.log 2>&1
cat $OUT/list_sweep.log
# (3) the reference's 32 MiB protocol: two dependent launches against the sequential single launch
timeout 600 $S \
  1024x1024 complex64 0.03125 auto,seq \
  1024x1024 complex128 0.03125 auto,seq \
  128x128x128 complex64 0.03125 auto,seq \
  128x128x128 complex128 0.03125 auto,seq \
  1048576 complex64 0.03125 auto,seq \
  262144 complex64 0.03125 auto,seq \
  1024x1024 complex64 0.125 auto,seq \
  128x128x128 complex64 0.125 auto,seq \
  1048576 complex64 0.125 auto,seq \
  1024x1024 complex64 0.25 auto,seq \
  > $OUT/small_sweep.log 2>&1
cat $OUT/small_sweep.log
# (4) where a 1 GiB execute spends its time: kernel durations against the per-execute wall time
rocprofv3 --kernel-trace --stats -d $OUT/trace1g -o t -- python3 tools/fused_sweep.py 1048576 complex64 1 auto > $OUT/trace1g.log 2>&1
find $OUT/trace1g -name "*kernel_stats.csv" | head -1 | xargs -r head -8
tail -2 $OUT/trace1g.log

#!/bin/bash
# round 4, call l: pair stores with anchored inter-pass twiddles (C4, cubes); fp64 2^16 ... 2^18 on the persistent kernel
set -u
OUT=gpurun_out/r04l
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python3 -m pytest tests/test_round3_gpu.py tests/test_round4_gpu.py tests/test_full_size_gpu.py tests/test_errors_gpu.py -q -m gpu -k "pair or cube or 256 or c4 or config4 or 3d or smooth_3d" > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
S="python3 tools/fused_sweep.py"
timeout 900 $S \
  128x128x128 complex64 4 pipelined,auto,auto@MIFFT_PAIR=6 \
  128x128x128 complex128 4 pipelined,auto \
  128x128x128 complex64 1 auto \
  128x128x128 complex128 1 auto \
  65536  complex128 1 pipelined,f:56:112,f:28:56 \
  65536  complex128 4 pipelined,f:56:112,f:28:56 \
  131072 complex128 1 pipelined,auto,f:28:56,f:14:28 \
  131072 complex128 4 pipelined,auto,f:28:56,f:14:28 \
  262144 complex128 1 pipelined,auto,f:14:28 \
  262144 complex128 4 pipelined,auto,f:14:28 \
  > $OUT/sweep.log 2>&1
cat $OUT/sweep.log
for c in c4 c4s; do timeout 600 python3 bench.py --config $c --steps 10 --warmup 2 > $OUT/bench_$c.json 2> $OUT/bench_$c.err; python3 -c "
import json; j=json.load(open('$OUT/bench_$c.json')); print('$c', j['roofline']['frac'], j['ms_per_step'], j['protocol']['out_of_place']['frac_median'], j['protocol']['in_place']['frac_median'], j['parity'])"; done

#!/bin/bash
# round 4, call k: the fp32 cube with a radix-32 y stage (four exchanges per point); kernel stats of the plain pair launches
set -u
OUT=gpurun_out/r04k
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
S="python3 tools/fused_sweep.py"
timeout 900 $S \
  128x128x128 complex64 1 pipelined,auto,auto@MIFFT_PAIR=6 \
  128x128x128 complex64 4 pipelined,auto,auto@MIFFT_PAIR=6 \
  > $OUT/cube_sweep.log 2>&1
cat $OUT/cube_sweep.log
PYFFT_AMD_STRATEGY=pipelined rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pipe_cube -- python3 bench.py --config cube --plain --steps 10 --warmup 2 > $OUT/pipe_cube.json 2> $OUT/pipe_cube.err
find $OUT/pipe_cube -name "*kernel_stats.csv" | head -1 | xargs -r head -4 | cut -c1-260
timeout 900 python3 -m pytest tests/test_round2_gpu.py tests/test_round3_gpu.py tests/test_round4_gpu.py -q -m gpu -k "any_size or mixed or smooth or bluestein" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log

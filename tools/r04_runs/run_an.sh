#!/bin/bash
# round 4, run an: split-complex fp32 1-D with the sibling tiles interleaved at lane level (whole lines per wave instruction)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04an
timeout 900 python -m pytest tests/test_round4_gpu.py tests/test_round2_gpu.py tests/test_errors_gpu.py -q -x -k "split or strateg or fused" 2>&1 | tail -5 > gpurun_out/r04an/tests.log
timeout 900 python tools/fused_sweep.py 1048576 float32 2 auto,f:7:14,auto@MIFFT_STORE=1 524288 float32 2 auto,auto@MIFFT_STORE=1 262144 float32 2 auto,f:28:56,f:28:56@MIFFT_STORE=1 \
   131072 float32 2 auto,auto@MIFFT_STORE=1 65536 float32 2 auto,auto@MIFFT_STORE=1 1048576 complex64 2 auto 1024x1024 float32 2 auto > gpurun_out/r04an/sweep.log 2>&1
timeout 600 python3 tools/pmc_traffic.py --tag r04an c2s > gpurun_out/r04an/pmc.log 2>&1
timeout 300 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --repeats 3 > gpurun_out/r04an/bench_c2.json 2>/dev/null
cat gpurun_out/r04an/tests.log; cat gpurun_out/r04an/sweep.log; tail -2 gpurun_out/r04an/pmc.log; python3 -c "
import json; j=json.load(open('gpurun_out/r04an/bench_c2.json')); print('c2', j['roofline']['frac'], j['protocol']['out_of_place']['frac_median'])"

#!/bin/bash
# round 4, run bd: PMC traffic + kernel stats of configuration 3 in the split layout (row-first kernel), then its bench line with the traffic
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04bd
timeout 800 python3 tools/pmc_traffic.py --tag r04 c3s c2s > gpurun_out/r04bd/pmc.log 2>&1
timeout 400 python3 bench.py --config c3s --steps 10 --warmup 2 > gpurun_out/r04bd/bench_c3s.json 2> gpurun_out/r04bd/bench_c3s.err
timeout 400 python3 bench.py --config c2s --steps 10 --warmup 2 > gpurun_out/r04bd/bench_c2s.json 2> gpurun_out/r04bd/bench_c2s.err
cp profiles/traffic_c3s.json profiles/traffic_c2s.json profiles/r04_c3s_kernel_stats.csv profiles/r04_c2s_kernel_stats.csv gpurun_out/r04bd/ 2>/dev/null
tail -3 gpurun_out/r04bd/pmc.log

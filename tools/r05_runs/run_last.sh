#!/bin/bash
# round 5, final tree: whole GPU suite (default + development library), then sections 1-2 of tools/make_profiles.sh and the reference's tables
set -u
TAG=r05
OUT=gpurun_out/r05last
mkdir -p $OUT/to_profiles
P=$OUT/to_profiles
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > $OUT/tests_gpu.log 2>&1
echo "tests rc=$?" >> $OUT/tests_gpu.log
tail -12 $OUT/tests_gpu.log
PYFFT_AMD_DEV_BUILD=1 timeout 900 python -m pytest tests -m gpu -x -q -k "xcd2 or per_xcd or sequential or wide_tiles or alternating_counter or fused_2d_split_row_first or direct_abi" > $OUT/tests_gpu_dev.log 2>&1
echo "tests rc=$?" >> $OUT/tests_gpu_dev.log
tail -3 $OUT/tests_gpu_dev.log
timeout 2700 python3 tools/pmc_traffic.py --tag $TAG c2 c3 c4 c4s c5 cube cubed c2s c3s > $OUT/pmc_traffic.log 2>&1
cp $OUT/pmc_traffic.log $P/${TAG}_pmc_traffic.log
cp profiles/traffic_*.json profiles/${TAG}_*_kernel_stats.csv $P/ 2>/dev/null
timeout 600 python3 bench.py > $OUT/bench_c2.json 2> $OUT/bench_c2.err
for c in c1 c3 c4 c4s cube cubed c2s c3s; do
    timeout 600 python3 bench.py --config $c --steps 10 --warmup 2 > $OUT/bench_$c.json 2> $OUT/bench_$c.err
done
timeout 900 python3 bench.py --config c5 --warmup 2 > $OUT/bench_c5.json 2> $OUT/bench_c5.err
timeout 600 python3 bench.py --config c5 --chunk-only --steps 10 --warmup 2 > $OUT/bench_c5chunk.json 2> $OUT/bench_c5chunk.err
for c in c1 c2 c3 c4 c4s c5 c5chunk cube cubed c2s c3s; do cp $OUT/bench_$c.json $P/${TAG}_bench_$c.json; done
timeout 900 python3 tools/perf_table.py > $OUT/perf_table.log 2>&1 && cp $OUT/perf_table.log $P/${TAG}_perf_table_reference_shapes.log
timeout 900 python3 tools/quick_bench.py tail 2>&1 | sed 's/passes=\[.*\]//' > $OUT/tail.log; cp $OUT/tail.log $P/${TAG}_tail_survey.log
cat $OUT/pmc_traffic.log; head -c 300 $OUT/bench_c2.json; echo

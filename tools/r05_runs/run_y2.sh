#!/bin/bash
# round 5, final library: sections 1-2 of tools/make_profiles.sh again (PMC traffic + kernel stats, every bench line), the reference's tables
set -u
TAG=r05
OUT=gpurun_out/r05y2
mkdir -p $OUT/to_profiles
P=$OUT/to_profiles
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
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
timeout 300 python3 tools/small_batch_probe.py sp > $OUT/small_batch.log 2>&1
timeout 300 python3 tools/small_batch_probe.py dp >> $OUT/small_batch.log 2>&1
cp $OUT/small_batch.log $P/${TAG}_small_batch_32MiB.log
timeout 900 python3 tools/quick_bench.py r5 2>&1 | sed 's/passes=\[.*\]//' > $OUT/r5_shapes.log; cp $OUT/r5_shapes.log $P/${TAG}_round5_shapes.log
cat $OUT/pmc_traffic.log; head -c 600 $OUT/bench_c2.json; echo; cat $OUT/small_batch.log | cut -c1-120

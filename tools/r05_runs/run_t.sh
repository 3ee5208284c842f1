#!/bin/bash
# round 5, end: refresh the shape tables that the later kernel work moved (the bench lines of the BASELINE configurations are unaffected)
set -u
OUT=gpurun_out/r05t
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python3 tools/quick_bench.py r5 2>&1 | sed 's/passes=\[.*\]//' > $OUT/r05_round5_shapes.log
timeout 900 python3 tools/quick_bench.py tail 2>&1 | sed 's/passes=\[.*\]//' > $OUT/r05_tail_survey.log
timeout 900 python3 tools/perf_table.py > $OUT/r05_perf_table_reference_shapes.log 2>&1
timeout 300 python3 tools/small_batch_probe.py sp > $OUT/r05_small_batch_32MiB.log 2>&1
timeout 300 python3 tools/small_batch_probe.py dp >> $OUT/r05_small_batch_32MiB.log 2>&1
timeout 900 python3 tools/quick_bench.py huge 2>&1 | sed 's/passes=\[.*\]//' > $OUT/r05_one_tile_shapes.log
timeout 600 python3 bench.py > $OUT/bench_c2.json 2> $OUT/bench_c2.err
cat $OUT/r05_perf_table_reference_shapes.log | head -14; cat $OUT/r05_small_batch_32MiB.log | cut -c1-110; cut -c1-300 $OUT/bench_c2.json

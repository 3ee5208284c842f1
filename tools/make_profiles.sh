#!/bin/bash
# Regenerates the round's evidence under profiles/ ON THE GPU BOX (run through gpurun from the repo root):
#     git rev-parse HEAD > .tree_commit; gpurun --timeout 3000 -- 'bash tools/make_profiles.sh r06'
# Everything is written to gpurun_out/<tag>/ (scratch, merged back by gpurun); the summaries that are judged are gathered in
# gpurun_out/<tag>/to_profiles/ under their final names -- back in the build container:
#     cp gpurun_out/<tag>/to_profiles/* profiles/;  rocprofv3 gets the program directly after `--` (no env/bash hop), counters in their own runs.
set -u
TAG=${1:-r06}
OUT=gpurun_out/$TAG
mkdir -p $OUT/to_profiles profiles
P=$OUT/to_profiles
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"

# 1. kernel stats + PMC traffic of the same command, per configuration (writes profiles/traffic_<c>.json and
#    profiles/<tag>_<c>_kernel_stats.csv)
timeout 2400 python3 tools/pmc_traffic.py --tag $TAG c2 c3 c4 c4s c5 cube cubed c2s c3s > $OUT/pmc_traffic.log 2>&1
cp $OUT/pmc_traffic.log $P/${TAG}_pmc_traffic.log
cp profiles/traffic_*.json profiles/${TAG}_*_kernel_stats.csv $P/ 2>/dev/null

# 2. the bench lines (after the traffic files exist, so that every line carries roofline.traffic): default line (c2) and every other BASELINE configuration
timeout 600 python3 bench.py > $OUT/bench_c2.json 2> $OUT/bench_c2.err
# (round 6: no --steps: K = max(10, 250 ms / step) after an untimed spin-up, so that short steps are not measured on the clock ramp)
for c in c1 c3 c4 c4s cube cubed c2s c3s; do
    timeout 600 python3 bench.py --config $c > $OUT/bench_$c.json 2> $OUT/bench_$c.err
done
# configuration 5 as BASELINE.json states it: the per-GPU share of 8192 transforms resident (256 GiB), 32 chunk executes = one sweep; and the
# rounds 1-4 form (one resident chunk, out of place) next to it
timeout 900 python3 bench.py --config c5 --warmup 2 > $OUT/bench_c5.json 2> $OUT/bench_c5.err
timeout 600 python3 bench.py --config c5 --chunk-only > $OUT/bench_c5chunk.json 2> $OUT/bench_c5chunk.err
for c in c1 c2 c3 c4 c4s c5 c5chunk cube cubed c2s c3s; do cp $OUT/bench_$c.json $P/${TAG}_bench_$c.json; done

# 3. the reference's published shapes (test/test_performance.py method), the vendor yardstick (cuda/test.cu counterpart) and its
#    value cross-check
timeout 900 python3 tools/perf_table.py > $OUT/perf_table.log 2>&1 && cp $OUT/perf_table.log $P/${TAG}_perf_table_reference_shapes.log
[ -x tools/rocfft_compare ] && timeout 600 ./tools/rocfft_compare > $OUT/rocfft.log 2>&1 && cp $OUT/rocfft.log $P/${TAG}_rocfft_comparator.log
timeout 600 python3 tools/hipfft_check.py > $OUT/hipfft_check.log 2>&1; cp $OUT/hipfft_check.log $P/${TAG}_hipfft_value_check.log

# 4. memory-system ceilings
[ -x tools/membench ] && timeout 300 ./tools/membench > $OUT/membench.log 2>&1 && cp $OUT/membench.log $P/${TAG}_fabric_ceiling_membench.log

# 5. the long 1-D sizes in both precisions, the small-batch rows of the reference's 32 MiB protocol
timeout 900 python3 tools/quick_bench.py 1d > $OUT/long_1d.log 2>&1
timeout 900 python3 tools/quick_bench.py f64 >> $OUT/long_1d.log 2>&1
echo "# the two-pass fp32 sizes at 1 GiB and at 8 GiB per side" >> $OUT/long_1d.log
timeout 900 python3 tools/quick_bench.py 1d1g >> $OUT/long_1d.log 2>&1
timeout 900 python3 tools/quick_bench.py 1d8g >> $OUT/long_1d.log 2>&1
cp $OUT/long_1d.log $P/${TAG}_long_1d_sizes.log
timeout 900 python3 tools/quick_bench.py r4 > $OUT/r4_shapes.log 2>&1; cp $OUT/r4_shapes.log $P/${TAG}_cubes_rectangles_fp64.log
# (second batch of round 4: 2-D shapes with a 256-point axis, {64, 128}^3, split-complex layouts -- 2 GiB per side)
timeout 900 python3 tools/quick_bench.py r4b 2>&1 | sed 's/passes=\[.*\]//' > $OUT/r4b_shapes.log; cp $OUT/r4b_shapes.log $P/${TAG}_second_batch_shapes.log
timeout 600 python3 tools/mixed_probe.py > $OUT/mixed.log 2>&1; cp $OUT/mixed.log $P/${TAG}_mixed_radix.log
timeout 300 python3 tools/small_batch_probe.py sp > $OUT/small_batch.log 2>&1
timeout 300 python3 tools/small_batch_probe.py dp >> $OUT/small_batch.log 2>&1
cp $OUT/small_batch.log $P/${TAG}_small_batch_32MiB.log

# 5b. round 5: pass-pair chains and the kernels with several work-groups per transform; the tail survey again
timeout 900 python3 tools/quick_bench.py r5 2>&1 | sed 's/passes=\[.*\]//' > $OUT/r5_shapes.log; cp $OUT/r5_shapes.log $P/${TAG}_round5_shapes.log
timeout 900 python3 tools/quick_bench.py tail 2>&1 | sed 's/passes=\[.*\]//' > $OUT/tail.log; cp $OUT/tail.log $P/${TAG}_tail_survey.log
# 5c. the planner's tuning table re-measured (every rule's persistent launch against the pipelined chunks, 2 GiB per side)
timeout 1500 python3 tools/fused_sweep.py --emit $OUT/tuning_emitted.json --gib 2 > $OUT/tuning_emit.log 2>&1
cp $OUT/tuning_emit.log $P/${TAG}_tuning_emit.log; cp $OUT/tuning_emitted.json $P/${TAG}_tuning_emitted.json

# 5d. round 6: SQ / TCP / TCC counters of the persistent kernels of C2 and C5 (one rocprofv3 --pmc run per counter group); the dense
#     split-complex N-D kernel against what ran before it; the composed upper bounds for the three-launch shapes; one process, four shards
timeout 1500 python3 tools/persistent_counters.py > $OUT/fused3_counters.log 2>&1; cp $OUT/fused3_counters.log $P/${TAG}_fused3_counters_final.log
timeout 900 python3 tools/planes_probe.py > $OUT/planes_probe.log 2>&1; cp $OUT/planes_probe.log $P/${TAG}_planes_probe_final.log
timeout 600 python3 tools/three_launch_probe.py > $OUT/three_launch.log 2>&1; cp $OUT/three_launch.log $P/${TAG}_three_launch_fused_final.log
timeout 900 python3 tools/plane_fused_probe.py > $OUT/plane_fused.log 2>&1; cp $OUT/plane_fused.log $P/${TAG}_plane_fused_probe_final.log
timeout 600 python3 bench.py --gpus 4 --single-process --share-gpu --no-cpu-baseline > $OUT/bench_c2_single_process_4.json 2> $OUT/bench_c2_single_process_4.err
cp $OUT/bench_c2_single_process_4.json $P/${TAG}_bench_c2_single_process_4_shards_one_gpu.json

# 6. SQ counters of the long fp32 rows (occupancy / LDS pressure)
timeout 900 python3 tools/row_counters.py 32768 complex64 8192 16384 complex64 16384 8192 complex64 32768 > $OUT/row_counters.log 2>&1
cp $OUT/row_counters.log $P/${TAG}_i_row_sq_counters.log
echo done

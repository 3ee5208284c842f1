#!/bin/bash
# gpurun -- 'bash tools/r03_runs/run_r03c.sh'
set -u
OUT=gpurun_out/r03c
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1500 python3 -m pytest tests/test_round3_gpu.py -x -q -m gpu -k "not error_grid" > $OUT/pytest_r3.log 2>&1
tail -5 $OUT/pytest_r3.log
for st in 0 2 1; do
  MIFFT_STORE=$st timeout 400 python3 tools/quick_bench.py ab > $OUT/ab_store$st.log 2>&1
done
paste -d'\n' $OUT/ab_store0.log $OUT/ab_store2.log $OUT/ab_store1.log | grep -v "^AMD" | awk '{print $1,$2,$4, $(NF-9), $(NF-8), $(NF-5), $(NF-4), $(NF-3), $(NF-2)}'
timeout 300 python3 tools/small_batch_probe.py sp > $OUT/small_sp.log 2>&1
timeout 300 python3 tools/small_batch_probe.py dp > $OUT/small_dp.log 2>&1
cat $OUT/small_sp.log $OUT/small_dp.log
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/small_prof -o small -- python3 tools/small_batch_probe.py sp > $OUT/small_prof.log 2>&1
find $OUT/small_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/small_kernel_stats.csv
rm -rf $OUT/small_prof
cut -c1-200 $OUT/small_kernel_stats.csv | head -30

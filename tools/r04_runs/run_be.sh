#!/bin/bash
# round 4, run be: rocprofv3 kernel statistics of the second-batch shape table (which kernels run, how long)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04be
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04be/prof -o r4b -- python3 tools/quick_bench.py r4b > gpurun_out/r04be/r4b.log 2>&1
f=$(find gpurun_out/r04be/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/r04be/r04_second_batch_kernel_stats.csv
head -30 gpurun_out/r04be/r04_second_batch_kernel_stats.csv | cut -c1-200

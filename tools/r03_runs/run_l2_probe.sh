#!/bin/bash
# gpurun -- 'bash tools/r03_runs/run_l2_probe.sh'   -> gpurun_out/r03a/{plain.log,pmc.log}
set -u
OUT=gpurun_out/r03a
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 300 ./tools/l2_resident_probe > $OUT/plain.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- ./tools/l2_resident_probe > $OUT/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- ./tools/l2_resident_probe > $OUT/write.log 2>&1
python3 tools/pmc_dispatch.py $OUT/fetch $OUT/write > $OUT/pmc.log 2>&1
rm -rf $OUT/fetch $OUT/write
cat $OUT/plain.log; head -150 $OUT/pmc.log

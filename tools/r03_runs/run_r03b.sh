#!/bin/bash
# gpurun -- 'bash tools/r03_runs/run_r03b.sh'
set -u
OUT=gpurun_out/r03b
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 300 python3 tools/pair_probe.py 16 > $OUT/pair_default.log 2>&1
MIFFT_PAIR=2 timeout 300 python3 tools/pair_probe.py 16 > $OUT/pair_alt.log 2>&1
MIFFT_PAIR=1 timeout 300 python3 tools/pair_probe.py 16 > $OUT/pair_off.log 2>&1
cat $OUT/pair_default.log $OUT/pair_alt.log $OUT/pair_off.log
timeout 120 ./tools/l2_resident_probe > $OUT/l2_plain.log 2>&1
cat $OUT/l2_plain.log
if grep -q "footprint  3072" $OUT/l2_plain.log; then
  timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- ./tools/l2_resident_probe > $OUT/fetch.log 2>&1
  timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- ./tools/l2_resident_probe > $OUT/write.log 2>&1
  python3 tools/pmc_dispatch.py $OUT/fetch $OUT/write > $OUT/l2_pmc.log 2>&1
  rm -rf $OUT/fetch $OUT/write
  cat $OUT/l2_pmc.log
fi

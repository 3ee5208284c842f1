#!/bin/bash
set -u
OUT=gpurun_out/r03i
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
for lr in "3 5" "4 6" "4 8" "5 10" "6 12" "8 16"; do set -- $lr
  for n in 19 20; do python3 tools/fusedx_probe.py one $n $1 $2 0; done
done > $OUT/sweep.log 2>&1
cat $OUT/sweep.log
for cnt in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $cnt --kernel-trace --output-format csv -d $OUT/pmc_$cnt -- python3 tools/fusedx_probe.py one 19 4 8 0 > $OUT/pmc_$cnt.log 2>&1
done
python3 tools/pmc_dispatch.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE | grep fused2x | tail -3
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE

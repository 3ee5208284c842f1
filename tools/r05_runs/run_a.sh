#!/bin/bash
# round 5, first GPU call (DEV build of the library): the round's new GPU tests, the anti-phase experiment on the XCD-resident kernel,
# configuration 5 as stated (bench line)
set -u
OUT=gpurun_out/r05a
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1500 python -m pytest tests/test_round5_gpu.py -x -q --durations=25 > $OUT/tests_round5.log 2>&1
echo "tests rc=$?" >> $OUT/tests_round5.log
timeout 900 python tools/xcd2_antiphase.py 512 2048 > $OUT/xcd2_antiphase.log 2>&1
echo "rc=$?" >> $OUT/xcd2_antiphase.log
timeout 900 python bench.py --config c5 > $OUT/bench_c5.json 2> $OUT/bench_c5.err
echo "rc=$?" >> $OUT/bench_c5.err
tail -5 $OUT/tests_round5.log; tail -30 $OUT/xcd2_antiphase.log; cat $OUT/bench_c5.json | cut -c1-1500; tail -5 $OUT/bench_c5.err

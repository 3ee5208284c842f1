#!/bin/bash
# round 4, call i: full GPU suite on the final tree + the mixed-radix probe with the fixed N-D tile capacity
set -u
OUT=gpurun_out/r04i
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1800 python3 -m pytest tests -q -m gpu > $OUT/pytest_all.log 2>&1; tail -8 $OUT/pytest_all.log
timeout 600 python3 tools/mixed_probe.py > $OUT/mixed.log 2>&1; tail -16 $OUT/mixed.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
timeout 600 python3 bench.py > $OUT/bench_c2.json 2> $OUT/bench_c2.err; cut -c1-400 $OUT/bench_c2.json

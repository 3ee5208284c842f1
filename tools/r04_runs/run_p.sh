#!/bin/bash
# round 4, call p: full GPU suite on the tree with the 32-column fp32 tiles as a default
set -u
OUT=gpurun_out/r04p
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1800 python3 -m pytest tests -q -m gpu > $OUT/pytest_all.log 2>&1; tail -8 $OUT/pytest_all.log
S="python3 tools/fused_sweep.py"
timeout 600 $S \
  262144 complex64 0.3 auto,pipelined \
  262144 complex64 0.5 auto,pipelined \
  131072 complex64 0.3 auto,pipelined \
  131072 complex64 0.5 auto,pipelined \
  65536 complex64 0.3 auto,pipelined \
  65536 complex64 0.5 auto,pipelined \
  > $OUT/sweep.log 2>&1
cat $OUT/sweep.log

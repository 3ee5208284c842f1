#!/bin/bash
# gpurun -- 'bash tools/r03_runs/run_r03d.sh'
set -u
OUT=gpurun_out/r03d
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python3 -m pytest tests/test_round3_gpu.py -x -q -m gpu -k "pass_pairs" > $OUT/pytest_pairs.log 2>&1
tail -5 $OUT/pytest_pairs.log
for c in c4 c4s; do
for v in "0 0" "0 3" "1 0"; do set -- $v
  MIFFT_PAIR=$1 MIFFT_STORE=$2 timeout 400 python3 bench.py --config $c --plain --steps 5 --warmup 2 > $OUT/${c}_pair$1_store$2.json 2> $OUT/${c}_pair$1_store$2.err
  python3 -c "import json,sys; d=json.load(open('$OUT/${c}_pair$1_store$2.json')); print('$c pair=$1 store=$2', d['ms_per_step'], d['roofline']['frac'], d['config']['strategy'], d['parity'])"
done
done
timeout 600 python3 tools/xcd2_elim.py 512 > $OUT/xcd2_elim.log 2>&1
cat $OUT/xcd2_elim.log
timeout 1500 python3 tools/small_fused_probe.py > $OUT/small_fused.log 2>&1
cat $OUT/small_fused.log

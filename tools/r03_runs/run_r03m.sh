#!/bin/bash
set -u
OUT=gpurun_out/r03m
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rep in 1 2; do for v in 0 3 4; do
  MIFFT_PAIR=$v timeout 400 python3 bench.py --config c4 --plain --steps 6 --warmup 2 > $OUT/c4_$v.json 2> $OUT/c4_$v.err
  python3 -c "import json; d=json.load(open('$OUT/c4_$v.json')); print('c4 MIFFT_PAIR=$v rep $rep', round(d['ms_per_step'],3), round(d['roofline']['frac'],4), d['parity']['ok'])"
done; done

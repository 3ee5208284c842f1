#!/bin/bash
# round 6: persistent kernels -- atomic optimizer off (tickets really a tile ahead), fused3p (next tile's loads first at the loop head)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x --durations=15 > $O/d_suite.log 2>&1; echo "suite rc=$?" >> $O/d_suite.log; tail -25 $O/d_suite.log
for c in c2 c3 cube; do timeout 600 python bench.py --config $c --no-cpu-baseline > $O/d_bench_$c.json 2> $O/d_bench_$c.err; done
timeout 600 python bench.py --config c5 --chunk-only --no-cpu-baseline --steps 20 > $O/d_bench_c5chunk.json 2> $O/d_bench_c5chunk.err
MIFFT_NO_PREFETCH=1 timeout 600 python bench.py --config c5 --chunk-only --no-cpu-baseline --steps 20 > $O/d_bench_c5chunk_noprefetch.json 2> $O/d_bench_c5chunk_noprefetch.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06/d_bench_*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f, "steps", d["steps"], "ms", round(d["ms_per_step"],4), "frac", round(d["roofline"]["frac"],4), "proto", d["roofline"].get("frac_protocol_median"), d["config"]["strategy"])
    except Exception as e:
        print(f, "FAILED", e, open(f.replace(".json",".err")).read()[-800:])
PY
timeout 900 python3 tools/quick_bench.py 1d > $O/d_long_1d.log 2>&1; tail -12 $O/d_long_1d.log
timeout 900 python3 tools/quick_bench.py f64 > $O/d_long_f64.log 2>&1; tail -12 $O/d_long_f64.log

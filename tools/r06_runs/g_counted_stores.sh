#!/bin/bash
# round 6: write-through stores the compiler counts (8-byte: atomic-store builtin, 16-byte: raw buffer store with sc1), pass-pair anchors two
# groups ahead, dense planes kernel selected per shape and size: the whole GPU suite, then the bench lines and the shape sweeps
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q -x --durations=12 > $O/g_suite.log 2>&1; echo "suite rc=$?" >> $O/g_suite.log; tail -20 $O/g_suite.log
for c in c2 c3 c4 c4s cube cubed c2s c3s; do timeout 600 python bench.py --config $c --no-cpu-baseline > $O/g_bench_$c.json 2> $O/g_bench_$c.err; done
timeout 600 python bench.py --config c5 --chunk-only --no-cpu-baseline --steps 20 > $O/g_bench_c5chunk.json 2> $O/g_bench_c5chunk.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06/g_bench_*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f, "steps", d["steps"], "ms", round(d["ms_per_step"],4), "frac", round(d["roofline"]["frac"],4), "proto", d["roofline"].get("frac_protocol_median"), d["config"]["strategy"])
    except Exception as e:
        print(f, "FAILED", e, open(f.replace(".json",".err")).read()[-800:])
PY
timeout 900 python3 tools/quick_bench.py f64 > $O/g_long_f64.log 2>&1; grep "^(" $O/g_long_f64.log | cut -c1-40,100-200
timeout 900 python3 tools/quick_bench.py r4 2>&1 | sed 's/passes=\[.*\]//' > $O/g_r4.log; grep "^(" $O/g_r4.log | cut -c1-150

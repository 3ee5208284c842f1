#!/bin/bash
# round 6: table look-ups batched and issued one exchange round ahead (fft_col2.hpp / fft_col3.hpp): parity of the affected kernels, then
# the bench lines and the long 1-D sizes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_errors_gpu.py tests/test_full_size_gpu.py tests/test_round5_gpu.py -m gpu -q -x -k "not fixed_shape_nd_kernels and not eight_ranks" > $O/c_parity.log 2>&1; echo "parity rc=$?" >> $O/c_parity.log; tail -3 $O/c_parity.log
for c in c2 c3; do timeout 600 python bench.py --config $c --no-cpu-baseline > $O/c_bench_$c.json 2> $O/c_bench_$c.err; done
timeout 600 python bench.py --config c5 --chunk-only --no-cpu-baseline --steps 20 > $O/c_bench_c5chunk.json 2> $O/c_bench_c5chunk.err
timeout 600 python bench.py --config c4 --no-cpu-baseline --steps 10 > $O/c_bench_c4.json 2> $O/c_bench_c4.err
timeout 600 python bench.py --config c2 --single-process --gpus 4 --share-gpu --batch 256 --no-cpu-baseline > $O/c_bench_c2_sp4.json 2> $O/c_bench_c2_sp4.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06/c_bench_*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f, "steps", d["steps"], "ms", round(d["ms_per_step"],4), "frac", round(d["roofline"]["frac"],4), "proto", d["roofline"].get("frac_protocol_median"), d["config"]["strategy"], d.get("n_gpus"))
    except Exception as e:
        print(f, "FAILED", e, open(f.replace(".json",".err")).read()[-800:])
PY
timeout 900 python3 tools/quick_bench.py 1d > $O/c_long_1d.log 2>&1; tail -30 $O/c_long_1d.log

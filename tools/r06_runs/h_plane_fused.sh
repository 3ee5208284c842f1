#!/bin/bash
# round 6: (y, x) planes of big 3-D transforms on the persistent 2-D kernels (strategy fused2z): parity, the composed bound next to the
# real thing (tools/three_launch_probe.py cube), A/B against the slab route; and the A/B of the pass-pair anchors two groups ahead
# (build_prev/libmifft.so = the library with round 5's pair_store, loaded as the "dev" library) on C4 / C4 split / the cubes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_persistent_gpu.py -m gpu -q -x -k "plane_fused" --durations=10 > $O/h_tests.log 2>&1; tail -15 $O/h_tests.log
timeout 600 python3 tools/three_launch_probe.py cube > $O/h_cube512.log 2>&1; cat $O/h_cube512.log
python3 - > $O/h_plane_fused_ab.log 2>&1 <<'PY'
import os, sys, numpy
sys.path.insert(0, os.getcwd())
from pyfft_amd.hip import DeviceArray, Plan
from pyfft_amd import _native as N
def run(shape, dtype, batch, env):
    for k in ("PYFFT_AMD_NO_PLANE_FUSED", "PYFFT_AMD_STRATEGY"): os.environ.pop(k, None)
    os.environ.update(env)
    dt = numpy.dtype(dtype); split = dt.kind == "f"
    size = int(numpy.prod(shape)); csz = dt.itemsize * (2 if split else 1)
    n = size * batch
    a = [DeviceArray((n,), dt) for _ in range(2 if split else 1)]
    b = [DeviceArray((n,), dt) for _ in range(2 if split else 1)]
    for x in a: N.check(N.lib.mifft_memset(x.ptr, 0, x.nbytes, None))
    plan = Plan(shape, dtype=dtype)
    bi = (a + [None])[:2]; bo = (b + [None])[:2]
    plan.timed_execute(2, False, False, batch, bi, bo)
    best = min(plan.timed_execute(4, False, False, batch, bi, bo) / 4 for _ in range(3))
    return 2.0 * n * csz / (best * 1e-3) / 8e12, plan.strategy(batch)
for shape, dtype, batch in [((512, 512, 512), "complex64", 2), ((512, 512, 512), "complex64", 4), ((256, 512, 512), "complex64", 8),
                            ((64, 512, 512), "complex64", 16), ((64, 1024, 1024), "complex64", 4), ((16, 1024, 1024), "complex64", 16),
                            ((128, 1024, 512), "complex64", 8), ((64, 2048, 2048), "complex64", 2), ((128, 512, 256), "complex64", 16), ((64, 1024, 256), "complex64", 16),
                            ((512, 512, 512), "complex128", 2), ((64, 512, 512), "complex128", 8), ((64, 512, 256), "complex128", 16), ((16, 1024, 512), "complex128", 16),
                            ((64, 1024, 1024), "complex128", 2), ((64, 1024, 1024), "float64", 2), ((8, 1024, 1024), "float64", 16)]:
    f1, s1 = run(shape, dtype, batch, {})
    f0, s0 = run(shape, dtype, batch, {"PYFFT_AMD_NO_PLANE_FUSED": "1"})
    print("%-18s %-10s x %-3d  planes on the persistent kernel %.3f %s   before %.3f %s" % (shape, dtype, batch, f1, s1, f0, s0), flush=True)
PY
cat $O/h_plane_fused_ab.log
# pass-pair anchors: this library against the one with round 5's pair_store
cp pyfft_amd/libmifft_dev.so $O/.dev_keep.so 2>/dev/null; cp build_prev/libmifft.so pyfft_amd/libmifft_dev.so
for rep in 1 2; do for c in c4 c4s cube cubed; do
  timeout 600 python bench.py --config $c --no-cpu-baseline > $O/h_ab_${c}_new_$rep.json 2>/dev/null
  PYFFT_AMD_DEV_BUILD=1 timeout 600 python bench.py --config $c --no-cpu-baseline > $O/h_ab_${c}_prev_$rep.json 2>/dev/null
done; done
cp $O/.dev_keep.so pyfft_amd/libmifft_dev.so 2>/dev/null; rm -f $O/.dev_keep.so
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06/h_ab_*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split('/')[-1], "ms", round(d["ms_per_step"],4), "frac", round(d["roofline"]["frac"],4), "proto", round(d["roofline"].get("frac_protocol_median") or 0,4))
    except Exception as e: print(f, "FAILED", e)
PY

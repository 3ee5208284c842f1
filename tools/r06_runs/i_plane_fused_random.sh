#!/bin/bash
# round 6: the plane-fused route on RANDOM data (run h measured it on zero-filled buffers, which the persistent kernels run ~10 % faster on
# than on noise -- the other routes do not), with the z pass on 32-column tiles as an A/B (dev library); C4's two levels against the
# number of side streams
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06; mkdir -p $O
PYFFT_AMD_DEV_BUILD=1 timeout 900 python3 tools/plane_fused_probe.py > $O/i_plane_fused_probe.log 2>&1; cat $O/i_plane_fused_probe.log
for rep in 1 2 3; do for ns in 2 1 3; do
  PYFFT_AMD_PIPE_STREAMS=$ns timeout 600 python bench.py --config c4 --no-cpu-baseline > $O/i_c4_streams${ns}_$rep.json 2>/dev/null
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06/i_c4_*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split('/')[-1], "ms", round(d["ms_per_step"],4), "frac", round(d["roofline"]["frac"],4), "proto", round(d["roofline"].get("frac_protocol_median") or 0,4), d["config"]["strategy"])
    except Exception as e: print(f, "FAILED", e)
PY

"""Ring size of the persistent two-pass kernel for N < 2^20: PYFFT_AMD_FUSED_LAGF sweep (lag = LAGF * grid / (4 * gsize), ring = 2 lag).
python3 tools/fused_lag_probe.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from fusedx_probe import CHILD
for log2n in (17, 18, 19, 20):
    for lagf in (0, 7, 14, 28, 56):
        e = dict(os.environ)
        if lagf:
            e.update({"PYFFT_AMD_STRATEGY": "fused", "PYFFT_AMD_FUSED_LAGF": str(lagf)})
        r = subprocess.run([sys.executable, "-c", CHILD, str(log2n)], env=e, capture_output=True, text=True)
        print(("LAGF=%d" % lagf if lagf else "auto").ljust(10), r.stdout.strip() or r.stderr.strip()[-300:], flush=True)

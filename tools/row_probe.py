"""ROW-kernel sweep: single-pass 1-D sizes, 1 GiB buffers (development tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy
from quick_bench import run

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "both"
    lo = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    hi = int(sys.argv[3]) if len(sys.argv) > 3 else 14
    if which in ("f32", "both"):
        for k in range(lo, hi + 1):
            run((1 << k,), numpy.complex64, (1 << 27) >> k)
    if which in ("f64", "both"):
        for k in range(lo, min(hi, 14) + 1):
            run((1 << k,), numpy.complex128, (1 << 26) >> k)

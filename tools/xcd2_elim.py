"""Elimination table of the XCD-cooperative single-crossing kernel (csrc/fft_xcd2.hpp), C2-shaped batch: the same launch with a
part of the work removed (wrong results by construction, same remaining traffic).  python3 tools/xcd2_elim.py [batch]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
from pyfft_amd import _native as N

n = 1 << 20
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
MODES = [(0, "the kernel"), (1, "no exchange waits (flags still written)"), (2, "no exchange at all (no scratch traffic, no flags)"),
         (3, "HBM only: loads + stores, no butterflies / LDS / exchange"), (5, "butterflies + LDS only")]
os.environ["PYFFT_AMD_STRATEGY"] = "xcd"
from pyfft_amd.hip import Plan, DeviceArray
rng = numpy.random.default_rng(7)
host = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(numpy.complex64)
a = DeviceArray((B * n,), numpy.complex64)
for i in range(B):
    N.check(N.lib.mifft_memcpy_h2d(a.ptr + i * n * 8, host.ctypes.data, n * 8, None))
out = DeviceArray((B * n,), numpy.complex64)
print("N = 2^20 complex64, batch %d; per transform and XCD: 8 MiB in + 8 MiB out; 70 %% of 8 TB/s = %.1f us per transform and XCD" % (B, 16 * n / 5.6e12 * 8 * 1e6))
for mode, name in MODES:
    os.environ["PYFFT_AMD_XCD2_FLAGS"] = str(1 | (mode << 4))
    plan = Plan((n,), dtype=numpy.complex64, wait_for_finish=True)
    assert plan.strategy(B)[0] == "xcd2"
    try:
        plan.timed_execute(1, False, False, B, [a, None], [out, None])
        ms = min(plan.timed_execute(5, False, False, B, [a, None], [out, None]) / 5 for _ in range(3))
    except RuntimeError as e:
        print("mode %d %-58s FAILED %s" % (mode, name, str(e)[:80]), flush=True)
        continue
    per = ms * 1e3 / (B / 8.0)
    print("mode %d %-58s %7.3f ms  %6.2f us per transform and XCD  %5.1f %% of 8 TB/s (as if correct)" % (
        mode, name, ms, per, 16.0 * n * B / (ms * 1e-3) / 8e12 * 100), flush=True)
os.environ["PYFFT_AMD_STRATEGY"] = "fused"
plan = Plan((n,), dtype=numpy.complex64, wait_for_finish=True)
ms = min(plan.timed_execute(5, False, False, B, [a, None], [out, None]) / 5 for _ in range(3))
print("fused2 (two crossings) on the same data %46.3f ms  %5.1f %% of 8 TB/s" % (ms, 16.0 * n * B / (ms * 1e-3) / 8e12 * 100))

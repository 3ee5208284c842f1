"""One strategy, a few executes, for rocprofv3 --pmc runs: python3 tools/xcd2_pmc.py <strategy> [batch] [xcd2 flags]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
os.environ["PYFFT_AMD_STRATEGY"] = sys.argv[1]
if len(sys.argv) > 3:
    os.environ["PYFFT_AMD_XCD2_FLAGS"] = sys.argv[3]
from pyfft_amd.hip import Plan, DeviceArray
from pyfft_amd import _native as N
n = 1 << 20
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
a = DeviceArray((B * n,), numpy.complex64)
b = DeviceArray((B * n,), numpy.complex64)
N.check(N.lib.mifft_memset(a.ptr, 0, a.nbytes, None))
plan = Plan((n,), dtype=numpy.complex64, wait_for_finish=True)
print(plan.strategy(B))
for _ in range(3):
    plan.execute(a, b, batch=B)

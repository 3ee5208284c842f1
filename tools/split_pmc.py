import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
os.environ["PYFFT_AMD_STRATEGY"] = "chain"
from pyfft_amd.hip import Plan, DeviceArray
from pyfft_amd import _native as N
n, B = 1 << 20, 256
bufs = [DeviceArray((n * B,), numpy.float32) for _ in range(4)]
for b in bufs[:2]:
    N.check(N.lib.mifft_memset(b.ptr, 0, b.nbytes, None))
plan = Plan((n,), dtype=numpy.float32, wait_for_finish=True)
for _ in range(2):
    plan.execute(bufs[0], bufs[1], bufs[2], bufs[3], batch=B)

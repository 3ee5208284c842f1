"""Host cost of one asynchronous execute() (a 16-point plan, batch 1: the device work is negligible).  Development tool."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
from pyfft_amd.hip import Plan, DeviceArray, Stream
a = DeviceArray((16,), numpy.complex64); b = DeviceArray((16,), numpy.complex64)
s = Stream()
plan = Plan((16,), dtype=numpy.complex64, stream=s)
for n in (2000, 20000):
    plan.execute(a, b); s.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        plan.execute(a, b)
    t1 = time.perf_counter()
    s.synchronize()
    t2 = time.perf_counter()
    print("%d executes: %.2f us per call on the host (%.2f us with the final sync)" % (n, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))

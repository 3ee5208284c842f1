"""The multi-pass rows of the reference's 32 MiB table (test/test_performance.py:11,22-30) one by one: passes, strategy, time per
execute (10 back-to-back executes between two events, best of 5) and the fraction of the 8 TB/s roofline.
    [BUF_MB=32] python3 tools/small_batch_probe.py [sp|dp]            run it under `rocprofv3 --kernel-trace --stats` for per-kernel times"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
from pyfft_amd.hip import Plan, DeviceArray, Event

SHAPES = [(1024, 1024), (16, 16, 128), (32, 32, 128), (128, 128, 128), (128, 128), (8192,), (1024,), (16, 16, 16)]
double = len(sys.argv) > 1 and sys.argv[1] == "dp"
dtype = numpy.complex128 if double else numpy.complex64
for shape in SHAPES:
    size = int(numpy.prod(shape))
    batch = (int(os.environ.get("BUF_MB", "32")) << 20) // (size * numpy.dtype(dtype).itemsize)
    rng = numpy.random.default_rng(5)
    data = (rng.standard_normal(size * batch) + 1j * rng.standard_normal(size * batch)).astype(dtype)
    a = DeviceArray((size * batch,), dtype).set(data)
    b = DeviceArray((size * batch,), dtype)
    plan = Plan(shape if len(shape) > 1 else shape[0], dtype=dtype, wait_for_finish=True)
    plan.execute(a, b, batch=batch)
    st = plan._context.getQueue()
    best = 1e9
    for _ in range(5):
        e0 = Event().record(st)
        for _ in range(10):
            plan.execute(a, b, batch=batch, wait_for_finish=False)
        e1 = Event().record(st)
        e1.synchronize()
        best = min(best, e1.time_since(e0) / 10)
    plan.finish()
    frac = 2.0 * size * batch * numpy.dtype(dtype).itemsize / (best * 1e-3) / 8e12
    print("%-16s x %-5d %s  %7.2f us  %.3f of roofline  %-10s %s" % (shape, batch, "dp" if double else "sp", best * 1e3, frac,
                                                                    plan.strategy(batch)[0], plan.pass_list()), flush=True)

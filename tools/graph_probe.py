"""Does a hipGraph of a small plan's launch chain run faster than the launches themselves?  The multi-pass rows of the reference's
32 MiB table: 10 back-to-back executes against 10 launches of the captured graph (best of 5).  Development probe (ctypes on
libamdhip64: stream capture of the plan's own stream)."""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
from pyfft_amd.hip import Plan, DeviceArray, Event

hip = ctypes.CDLL("libamdhip64.so")
SHAPES = [(1024, 1024), (16, 16, 128), (32, 32, 128), (128, 128, 128), (128, 128), (1048576,)]
dtype = numpy.complex64
for shape in SHAPES:
    size = int(numpy.prod(shape))
    batch = (32 << 20) // (size * numpy.dtype(dtype).itemsize)
    a = DeviceArray((size * batch,), dtype).set(numpy.ones(size * batch, dtype))
    b = DeviceArray((size * batch,), dtype)
    plan = Plan(shape if len(shape) > 1 else shape[0], dtype=dtype, wait_for_finish=True)
    plan.execute(a, b, batch=batch)
    ref = b.get()
    st = plan._context.getQueue()
    sh = ctypes.c_void_p(plan._context.stream_handle())

    def timed(fn):
        best = 1e9
        for _ in range(5):
            e0 = Event().record(st)
            for _ in range(10):
                fn()
            e1 = Event().record(st)
            e1.synchronize()
            best = min(best, e1.time_since(e0) / 10)
        return best * 1e3
    t_plain = timed(lambda: plan.execute(a, b, batch=batch, wait_for_finish=False))
    graph, gexec = ctypes.c_void_p(), ctypes.c_void_p()
    rc = hip.hipStreamBeginCapture(sh, 0)
    plan.execute(a, b, batch=batch, wait_for_finish=False)
    rc2 = hip.hipStreamEndCapture(sh, ctypes.byref(graph))
    rc3 = hip.hipGraphInstantiate(ctypes.byref(gexec), graph, None, None, 0)
    if rc or rc2 or rc3:
        print(shape, "capture failed", rc, rc2, rc3)
        continue
    b.set(numpy.zeros(size * batch, dtype))
    hip.hipGraphLaunch(gexec, sh)
    plan.finish()
    same = numpy.array_equal(b.get(), ref)
    t_graph = timed(lambda: hip.hipGraphLaunch(gexec, sh))
    plan.finish()
    print("%-16s x %-5d %-10s launches %7.2f us   graph %7.2f us   identical %s" % (shape, batch, plan.strategy(batch)[0], t_plain, t_graph, same),
          flush=True)

"""debug: which exception does an unprepared captured execute raise, and what does end_capture answer (round 5)"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy
from pyfft_amd import hip, _native as N

for shape, batch, dtype in (((1 << 18,), 160, numpy.complex64), ((128, 128, 128), 32, numpy.complex64), ((256, 4096), 40, numpy.complex64)):
    size = int(numpy.prod(shape))
    s = hip.Stream()
    plan = hip.Plan(shape if len(shape) > 1 else shape[0], dtype=dtype, stream=s)
    a = hip.DeviceArray((size * batch,), dtype)
    b = hip.DeviceArray((size * batch,), dtype)
    N.check(N.lib.mifft_memset(a.ptr, 0, a.nbytes, s.handle))
    plan.execute(a, b, batch=batch)
    s.synchronize()
    print(shape, plan.strategy(batch), flush=True)
    with hip.Graph(s) as g:
        plan.execute(a, b, batch=batch)
    g.launch(); s.synchronize(); plan.finish()
    print("  replay ok", flush=True)
    N.check(N.lib.mifft_stream_begin_capture(s.handle))
    try:
        plan.execute(a, b, batch=batch - 1)
        print("  no exception?!")
    except Exception as e:
        print("  exception:", repr(e)[:300], flush=True)
    h = ctypes.c_void_p()
    rc = N.lib.mifft_stream_end_capture(s.handle, ctypes.byref(h))
    print("  end_capture rc", rc, N.last_error() if rc else "", "graph", h.value, flush=True)
    c = ctypes.c_int32()
    N.lib.mifft_stream_is_capturing(s.handle, ctypes.byref(c))
    print("  still capturing:", c.value, flush=True)
    try:
        s.synchronize(); print("  sync ok")
    except Exception as e:
        print("  sync:", repr(e)[:200])

"""debug: does freeing device memory during a capture invalidate it? (round 5)"""
import os, sys, ctypes, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy
from pyfft_amd import hip, _native as N

s = hip.Stream()
plan = hip.Plan(1 << 18, dtype=numpy.complex64, stream=s)
a = hip.DeviceArray(((1 << 18) * 160,), numpy.complex64)
N.check(N.lib.mifft_memset(a.ptr, 0, a.nbytes, s.handle))
plan.execute(a, batch=160); s.synchronize()
for what in ("free", "nothing"):
    victim = hip.DeviceArray((1 << 20,), numpy.complex64)
    N.check(N.lib.mifft_stream_begin_capture(s.handle))
    plan.execute(a, batch=160)
    if what == "free":
        del victim          # refcount -> __del__ -> hipFree inside the capture window
    h = ctypes.c_void_p()
    rc = N.lib.mifft_stream_end_capture(s.handle, ctypes.byref(h))
    c = ctypes.c_int32(); N.lib.mifft_stream_is_capturing(s.handle, ctypes.byref(c))
    print(what, "-> end_capture rc", rc, (N.last_error() if rc else ""), "still capturing", c.value, flush=True)
    try:
        s.synchronize(); print("   sync ok")
    except Exception as e:
        print("   sync:", repr(e)[:200])
        break

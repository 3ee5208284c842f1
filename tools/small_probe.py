"""Small-transform sweep (f3): wave-autonomous kernels (csrc/fft_wave.hpp) against the LDS-staged kernels they replace, at the
reference's 32 MiB buffer (test/helpers.py:7) and at 1 GiB.  Development tool; output kept as profiles/r02_*_small_n.log."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
from pyfft_amd.hip import Plan, DeviceArray, Event
from pyfft_amd import _native as N


def run(shape, dtype, buffer_mib, no_wave):
    N.check(N.lib.mifft_debug_set(N.DEBUG_NO_WAVE, 1 if no_wave else 0))
    N.check(N.lib.mifft_debug_set(N.DEBUG_FORCE_WAVE, 0 if no_wave else 1))
    size = int(numpy.prod(shape))
    isz = numpy.dtype(dtype).itemsize
    batch = (buffer_mib << 20) // (size * isz)
    a = DeviceArray((size * batch,), dtype)
    b = DeviceArray((size * batch,), dtype)
    rng = numpy.random.default_rng(3)
    blk = rng.standard_normal(1 << 22).astype(numpy.float32)      # 16 MiB of N(0,1) values, tiled over the buffer
    done = 0
    while done < a.nbytes:
        n = min(blk.nbytes, a.nbytes - done)
        N.check(N.lib.mifft_memcpy_h2d(a.ptr + done, blk.ctypes.data, n, None))
        done += n
    plan = Plan(shape, dtype=dtype, wait_for_finish=True)
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
    N.check(N.lib.mifft_debug_set(N.DEBUG_NO_WAVE, 0))
    N.check(N.lib.mifft_debug_set(N.DEBUG_FORCE_WAVE, 0))
    return 2.0 * size * batch * isz / (best * 1e-3) / 8e12, best * 1e3


if __name__ == "__main__":
    print("%-10s %-10s %6s | %-22s | %-22s" % ("shape", "dtype", "MiB", "LDS kernel  frac  us", "wave kernel frac  us"))
    for buffer_mib in (32, 128, 256, 1024):
        for shape, dtype in (((4,), numpy.complex64), ((8,), numpy.complex64), ((16,), numpy.complex64), ((32,), numpy.complex64),
                             ((4,), numpy.complex128), ((8,), numpy.complex128), ((16,), numpy.complex128), ((16, 16), numpy.complex64)):
            f0, t0 = run(shape, dtype, buffer_mib, True)
            f1, t1 = run(shape, dtype, buffer_mib, False)
            print("%-10s %-10s %6d | %10.3f %10.1f | %10.3f %10.1f" % (shape, numpy.dtype(dtype).name, buffer_mib, f0, t0, f1, t1), flush=True)

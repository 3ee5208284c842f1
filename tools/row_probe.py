"""Long-row sweep: plain vs persistent (prefetching) form of the register-edged ROW kernels, 1 GiB buffers, random data.
Also checks both forms against numpy on a few rows.  Development tool."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
from pyfft_amd.hip import Plan, DeviceArray, Event
from pyfft_amd import _native as N


def run(L, dtype, persist, alt=0):
    N.check(N.lib.mifft_debug_set(N.DEBUG_ALT_ROWS, alt))
    N.check(N.lib.mifft_debug_set(N.DEBUG_PERSIST, 1 if persist else 0))
    isz = numpy.dtype(dtype).itemsize
    batch = (1 << 30) // (L * isz)
    rng = numpy.random.default_rng(3)
    blk = (rng.standard_normal(4 * L) + 1j * rng.standard_normal(4 * L)).astype(dtype)
    a = DeviceArray((L * batch,), dtype)
    b = DeviceArray((L * batch,), dtype)
    done = 0
    while done < a.nbytes:
        n = min(blk.nbytes, a.nbytes - done)
        N.check(N.lib.mifft_memcpy_d2d(a.ptr + done, a.ptr, min(done, n), None)) if done >= blk.nbytes else \
            N.check(N.lib.mifft_memcpy_h2d(a.ptr + done, blk.ctypes.data, n, None))
        done += min(done, n) if done >= blk.nbytes else n
    plan = Plan((L,), dtype=dtype, wait_for_finish=True)
    plan.execute(a, b, batch=batch)
    got = numpy.empty(4 * L, dtype)
    N.check(N.lib.mifft_memcpy_d2h(got.ctypes.data, b.ptr + (batch - 4) * L * isz, got.nbytes, None))
    src = numpy.empty(4 * L, dtype)
    N.check(N.lib.mifft_memcpy_d2h(src.ctypes.data, a.ptr + (batch - 4) * L * isz, src.nbytes, None))
    ref = numpy.fft.fft(src.astype(numpy.complex128).reshape(4, L), axis=1).ravel()
    err = numpy.abs(got - ref).max() / numpy.abs(ref).max()
    st = plan._context.getQueue()
    best = 1e9
    for _ in range(3):
        e0 = Event().record(st)
        for _ in range(5):
            plan.execute(a, b, batch=batch, wait_for_finish=False)
        e1 = Event().record(st)
        e1.synchronize()
        best = min(best, e1.time_since(e0) / 5)
    N.check(N.lib.mifft_debug_set(N.DEBUG_PERSIST, 0))
    N.check(N.lib.mifft_debug_set(N.DEBUG_ALT_ROWS, 0))
    return 2.0 * L * batch * isz / (best * 1e-3) / 8e12, err


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "alt":
    for L in (16384, 32768):
        res = {}
        for rep in range(4):                      # interleaved repeats: the pool drifts by a few per cent within a run
            for alt in (0, 1, 2, 3):
                f, e = run(L, numpy.complex64, False, alt)
                res.setdefault(alt, []).append(f)
        for alt in sorted(res):
            print("L=%d stage-list variant %d: median %.3f  all %s" % (L, alt, sorted(res[alt])[len(res[alt]) // 2], ["%.3f" % x for x in res[alt]]), flush=True)
    sys.exit(0)
if __name__ == "__main__":
    print("%-8s %-10s | plain frac (err)      | persistent frac (err)" % ("L", "dtype"))
    for L, dtype in ((4096, numpy.complex64), (8192, numpy.complex64), (16384, numpy.complex64), (32768, numpy.complex64),
                     (4096, numpy.complex128), (8192, numpy.complex128), (16384, numpy.complex128)):
        f0, e0 = run(L, dtype, False)
        f1, e1 = run(L, dtype, True)
        print("%-8d %-10s | %.3f (%.1e)       | %.3f (%.1e)" % (L, numpy.dtype(dtype).name, f0, e0, f1, e1), flush=True)

"""Value cross-check against the vendor library (counterpart of the reference's cuda/test.cu:13-95, which runs CUFFT on the
published shapes): hipFFT/rocFFT and libmifft transform the SAME seeded device buffer, out of place, and the results are
compared with the reference's thresholds (test/test_errors.py:20-23: L1-relative difference < 1.1e-6 fp32 / 1e-11 fp64) and
the north star's max-norm bound; both are also compared with numpy.fft on the complex128-upcast input for the first
transform.  hipFFT is a third, independent checker here, never part of the product path.

    python3 tools/hipfft_check.py [--quick]      one line per shape, exit code 1 on a mismatch
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy

HIPFFT_C2C, HIPFFT_Z2Z, HIPFFT_FORWARD, HIPFFT_BACKWARD = 0x29, 0x69, -1, 1

# the comparator's shapes (tools/rocfft_compare.cpp) at batches that keep the host-side comparison short
SHAPES = [((1024,), 256, False), ((4096,), 64, False), ((1 << 16,), 16, False), ((1 << 20,), 4, False),
          ((1024, 1024), 4, False), ((256, 256, 256), 1, True), ((1 << 22,), 2, False), ((128, 128, 128), 4, False)]
QUICK = [((1024,), 64, False), ((64, 256), 8, True)]


def load_hipfft():
    for name in ("libhipfft.so", "/opt/rocm/lib/libhipfft.so", "libhipfft.so.0"):
        try:
            return ctypes.CDLL(name)
        except OSError:
            continue
    return None


def check(shape, batch, double, lib):
    from pyfft_amd.hip import Plan, DeviceArray
    dtype = numpy.complex128 if double else numpy.complex64
    fdt = numpy.float64 if double else numpy.float32
    size = int(numpy.prod(shape))
    rng = numpy.random.default_rng(20260 + size % 997)
    data = (rng.standard_normal(size * batch).astype(fdt) + 1j * rng.standard_normal(size * batch).astype(fdt)).astype(dtype)
    a = DeviceArray((size * batch,), dtype).set(data)
    ours = DeviceArray((size * batch,), dtype)
    theirs = DeviceArray((size * batch,), dtype)

    plan = Plan(shape if len(shape) > 1 else shape[0], dtype=dtype, wait_for_finish=True)
    plan.execute(a, ours, batch=batch)

    h = ctypes.c_void_p()
    dims = (ctypes.c_int * len(shape))(*shape)
    rc = lib.hipfftPlanMany(ctypes.byref(h), len(shape), dims, None, 1, size, None, 1, size,
                            HIPFFT_Z2Z if double else HIPFFT_C2C, batch)
    assert rc == 0, "hipfftPlanMany -> %d" % rc
    ex = lib.hipfftExecZ2Z if double else lib.hipfftExecC2C
    rc = ex(h, ctypes.c_void_p(a.ptr), ctypes.c_void_p(theirs.ptr), HIPFFT_FORWARD)
    assert rc == 0, "hipfftExec -> %d" % rc
    from pyfft_amd import _native as N
    N.check(N.lib.mifft_device_sync(), "sync")
    lib.hipfftDestroy(h)

    x = ours.get().astype(numpy.complex128)
    y = theirs.get().astype(numpy.complex128)
    assert numpy.array_equal(a.get(), data), "an out-of-place transform modified its input"
    diff = float(numpy.abs(x - y).sum() / numpy.abs(y).sum())
    mx = float(numpy.abs(x - y).max() / numpy.abs(y).max())
    ref = numpy.fft.fftn(data[:size].reshape(shape).astype(numpy.complex128)).reshape(-1)
    d_ours = float(numpy.abs(x[:size] - ref).sum() / numpy.abs(ref).sum())
    d_theirs = float(numpy.abs(y[:size] - ref).sum() / numpy.abs(ref).sum())
    eps, mtol = (1e-11, 1e-10) if double else (1.1e-6, 1e-5)
    ok = diff < eps and mx <= mtol and d_ours < eps and d_theirs < eps
    print("%-18s batch %-4d %s  |mifft - hipfft| L1-rel %.2e max-rel %.2e   vs numpy: mifft %.2e hipfft %.2e   %s" % (
        "x".join(map(str, shape)), batch, "z2z" if double else "c2c", diff, mx, d_ours, d_theirs, "ok" if ok else "MISMATCH"),
        flush=True)
    return ok


def main():
    lib = load_hipfft()
    if lib is None:
        print("hipFFT library not found")
        return 2
    lib.hipfftPlanMany.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                                   ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int,
                                   ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    for f in (lib.hipfftExecC2C, lib.hipfftExecZ2Z):
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    lib.hipfftDestroy.argtypes = [ctypes.c_void_p]
    shapes = QUICK if "--quick" in sys.argv else SHAPES
    ok = all([check(s, b, d, lib) for s, b, d in shapes])
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())

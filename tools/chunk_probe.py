"""Does the inter-pass scratch stay in the 256 MiB Infinity Cache when the batch is processed in
chunks?  Times N=2^20 c64, total batch B, executed as B/chunk calls sharing one temp buffer."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
from pyfft_amd.hip import Plan, DeviceArray, Event
from pyfft_amd import _native as N

def fill(b):
    rng = numpy.random.default_rng(1)
    host = rng.standard_normal(1 << 22).astype(numpy.float32)
    N.check(N.lib.mifft_memcpy_h2d(b.ptr, host.ctypes.data, host.nbytes, None))
    done = host.nbytes
    while done < b.nbytes:
        n = min(done, b.nbytes - done)
        N.check(N.lib.mifft_memcpy_d2d(b.ptr + done, b.ptr, n, None))
        done += n
    N.check(N.lib.mifft_device_sync())

def probe(shape, B, chunks, dtype=numpy.complex64, iters=3):
    size = int(numpy.prod(shape))
    nel = size * B
    a = DeviceArray((nel,), dtype); b = DeviceArray((nel,), dtype)
    fill(a)
    isz = numpy.dtype(dtype).itemsize
    for chunk in chunks:
        plan = Plan(shape, dtype=dtype, wait_for_finish=False)
        st = plan._context.getQueue()
        def once():
            for c in range(0, B, chunk):
                off = c * size * isz
                plan.execute(a.ptr + off, b.ptr + off, batch=chunk)
        once(); st.synchronize()
        best = 1e9
        for _ in range(iters):
            e0 = Event().record(st); once(); e1 = Event().record(st); e1.synchronize()
            best = min(best, e1.time_since(e0))
        alg = 2.0 * nel * isz
        print("%s B=%d chunk=%-4d (scratch %.0f MiB): %.3f ms  %.1f GB/s alg = %.1f%% of 8 TB/s" % (
            shape, B, chunk, chunk * size * isz / 2**20, best, alg / best / 1e6, alg / best / 1e6 / 80), flush=True)

if __name__ == "__main__":
    probe((1 << 20,), 256, [1, 2, 4, 8, 16, 32, 64, 256])
    probe((1024, 1024), 256, [1, 2, 4, 8, 16, 32, 256])
    probe((1 << 16,), 4096, [16, 64, 128, 256, 512, 4096])

"""Throughput of the opt-in extensions (pyfft_amd/generic.py): sizes that are not powers of two and tiled batches.
GFLOPS by the nominal 5*N*log2(N) count, and algorithmic GB/s (2 * N * sizeof(complex) per transform).  Development tool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
from pyfft_amd.hip import Plan, DeviceArray, Event
from pyfft_amd import _native as N

def run(shape, dtype, batch, parent=None):
    pshape = parent if parent else shape
    nel = batch * int(numpy.prod(pshape))
    split = numpy.dtype(dtype).kind == "f"                    # two scalar planes per side
    bufs = [DeviceArray((nel,), dtype) for _ in range(4 if split else 2)]
    for m in bufs:
        N.check(N.lib.mifft_memset(m.ptr, 0, m.nbytes, None))
    args = bufs if split else bufs[:2]
    plan = Plan(shape, dtype=dtype, any_size=True, parent_shape=parent, wait_for_finish=True)
    plan.execute(*args, batch=batch)
    st = plan._context.getQueue()
    e0 = Event().record(st)
    for _ in range(5):
        plan.execute(*args, batch=batch, wait_for_finish=False)
    e1 = Event().record(st); e1.synchronize()
    ms = e1.time_since(e0) / 5
    size = int(numpy.prod(shape)); ntr = nel // size
    fl = 5.0 * size * numpy.log2(size) * ntr
    print("%-16s parent %-14s %-10s transforms %-8d %.3f ms  %7.0f GFLOPS  %6.1f GB/s alg (%.1f%% of 8 TB/s)" % (
        shape, parent, numpy.dtype(dtype).name, ntr, ms, fl / ms / 1e6, 2.0 * nel * numpy.dtype(dtype).itemsize * (2 if split else 1) / ms / 1e6,
        2.0 * nel * numpy.dtype(dtype).itemsize * (2 if split else 1) / ms / 1e6 / 80), flush=True)

if __name__ == "__main__":
    c64, c128 = numpy.complex64, numpy.complex128
    if len(sys.argv) > 1 and sys.argv[1] == "tiled":          # tiles of a parent array, interleaved and split planes (round 4)
        for dt in (c64, numpy.float32, c128, numpy.float64):
            h = 2 if dt in (c128, numpy.float64) else 1      # the same bytes per side in both precisions
            run((128, 128), dt, 16 // h, parent=(4096, 4096)); run((64, 64), dt, 16 // h, parent=(4096, 4096))
            run((16, 16), dt, 16 // h, parent=(4096, 4096)); run((16, 16, 16), dt, 8 // h, parent=(256, 256, 256))
            run((8, 32, 32), dt, 8 // h, parent=(256, 256, 256))
        sys.exit(0)
    run((1000,), c64, 1 << 17); run((1024,), c64, 1 << 17); run((30000,), c64, 4096); run((1000,), c128, 1 << 16)
    run((100, 100), c64, 8192); run((60, 60, 60), c64, 512)
    run((128, 128), c64, 16, parent=(4096, 4096)); run((128, 128), c64, 16 * 1024)
    run((16, 16, 16), c64, 8, parent=(256, 256, 256))
    run((16, 16), c64, 16, parent=(4096, 4096)); run((64, 64), c64, 16, parent=(4096, 4096)); run((128, 128), c128, 8, parent=(4096, 4096))
    run((32, 32, 32), c64, 8, parent=(256, 256, 256)); run((8, 8, 8), c64, 8, parent=(256, 256, 256))

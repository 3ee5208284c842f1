"""Counterpart of the reference's test/test_performance.py: the 11 published shapes, batch chosen to fill a
32 MiB buffer (test_performance.py:11), warm-up + 10 timed out-of-place executes (median of 5 such blocks), GFLOPS by
5e-9 * (log2 x + log2 y + log2 z) * x*y*z * batch / t (test_performance.py:24).  Prints a table next to the
reference's published Tesla C2050 numbers (doc/source/index.rst:357-373)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
from pyfft_amd.hip import Plan, DeviceArray, Event
from pyfft_amd import _native as N

PUBLISHED_C2050 = {  # shape -> (pyfft sp, cufft sp, pyfft dp, cufft dp) GFLOPS
    (16,): (91.4, 127.1, 37.8, 28.3), (1024,): (254.0, 316.2, 28.4, 75.9), (8192,): (117.3, 250.6, 29.3, 94.7),
    (16, 16): (106.7, 119.7, 43.4, 39.4), (128, 128): (187.2, 198.7, 47.4, 53.4), (1024, 1024): (168.5, 184.4, 27.7, 73.7),
    (16, 16, 16): (117.6, 117.6, 47.0, 45.0), (32, 32, 128): (163.9, 161.6, 58.8, 63.2), (128, 128, 128): (184.7, 191.2, 44.6, 52.2),
}
SHAPES = [(16,), (1024,), (8192,), (16, 16), (128, 128), (1024, 1024), (8, 8, 64), (16, 16, 16), (16, 16, 128),
          (32, 32, 128), (128, 128, 128)]

def run(shape, double, buffer_mib, split=False):
    """split: the reference's other layout (dtype float32 / float64: re / im planes; the same bytes per side)"""
    dtype = numpy.complex128 if double else numpy.complex64
    size = int(numpy.prod(shape))
    batch = (buffer_mib << 20) // (size * numpy.dtype(dtype).itemsize)
    if batch == 0:
        return None
    rng = numpy.random.default_rng(5)
    fdt = numpy.float64 if double else numpy.float32
    if split:
        bufs = [DeviceArray((size * batch,), fdt).set(rng.standard_normal(size * batch).astype(fdt)) for _ in range(2)] + \
            [DeviceArray((size * batch,), fdt) for _ in range(2)]
        dtype = fdt
    else:
        data = (rng.standard_normal(size * batch).astype(fdt) + 1j * rng.standard_normal(size * batch).astype(fdt)).astype(dtype)
        bufs = [DeviceArray((size * batch,), dtype).set(data), DeviceArray((size * batch,), dtype)]
    plan = Plan(shape if len(shape) > 1 else shape[0], dtype=dtype, wait_for_finish=True)
    gflop = 5.0e-9 * sum(numpy.log2(s) for s in shape) * size * batch
    plan.execute(*bufs, batch=batch)
    st = plan._context.getQueue()
    # (round 6: the reference warms up with ONE execute; a device that idled while the host built the data set starts a 10 us kernel at
    # its idle clock -- the first row of a run read 82 us per execute instead of 10 --, so the warm-up is ~20 ms of executes here)
    import time
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.02:
        for _ in range(16):
            plan.execute(*bufs, batch=batch, wait_for_finish=False)
        plan.finish()
    # the reference times ONE block of 10 executes; a block is 100 us here and a single hiccup moves a row by a third, so: the median of 5
    ts = []
    for _ in range(5):
        e0 = Event().record(st)
        for _ in range(10):
            plan.execute(*bufs, batch=batch, wait_for_finish=False)
        e1 = Event().record(st); e1.synchronize()
        ts.append(e1.time_since(e0) / 1e3 / 10)
    t = sorted(ts)[2]
    return batch, t, gflop / t

if __name__ == "__main__":
    if "--quick" in sys.argv:
        # two shapes at the reference's 32 MiB buffer, machine-readable (tests/test_round3_gpu.py checks the GFLOPS formula)
        import json
        for shape in ((1024,), (128, 128)):
            batch, t, gf = run(shape, False, 32)
            print(json.dumps({"shape": list(shape), "batch": batch, "seconds_per_execute": t, "gflops": gf}), flush=True)
        sys.exit(0)
    if "--split" in sys.argv:
        # the same shapes in the reference's split-complex layout next to the interleaved one (second batch of round 4)
        for buffer_mib in (32, 1024):
            print("buffer %d MiB per side; GFLOPS interleaved / split planes, single and double precision" % buffer_mib)
            for shape in SHAPES:
                r = [run(shape, d, buffer_mib, sp)[2] for d in (False, True) for sp in (False, True)]
                print("%-16s sp %9.1f / %9.1f (%.2f)   dp %9.1f / %9.1f (%.2f)" % (str(shape), r[0], r[1], r[1] / r[0], r[2], r[3], r[3] / r[2]), flush=True)
        sys.exit(0)
    for buffer_mib in (32, 1024):
        print("buffer %d MiB (the reference uses 32 MiB, test/helpers.py:7)" % buffer_mib)
        print("%-16s %8s %12s %12s   %s" % ("shape", "batch", "sp GFLOPS", "dp GFLOPS", "published C2050 pyfft/cufft sp, pyfft/cufft dp"))
        for shape in SHAPES:
            sp = run(shape, False, buffer_mib); dp = run(shape, True, buffer_mib)
            pub = PUBLISHED_C2050.get(shape)
            print("%-16s %8d %12.1f %12.1f   %s" % (str(shape), sp[0], sp[2], dp[2], pub if pub else "-"), flush=True)

"""Quick per-shape timing sweep on one GPU (development tool, not the judged bench)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
from pyfft_amd.hip import Plan, DeviceArray, to_gpu, device_props

def run(shape, dtype, batch, iters=5, inplace=False):
    dt = numpy.dtype(dtype)
    size = int(numpy.prod(shape))
    split = dt.kind == 'f'
    itemsize = dt.itemsize * (2 if split else 1)
    rng = numpy.random.default_rng(1)
    nel = size * batch
    blk = min(nel, 1 << 22)
    host = (rng.standard_normal(blk * 2).astype(dt if split else (numpy.float32 if dt == numpy.complex64 else numpy.float64)))
    plan = Plan(shape, dtype=dtype)
    if split:
        bufs_in = [DeviceArray((nel,), dt), DeviceArray((nel,), dt)]
        bufs_out = bufs_in if inplace else [DeviceArray((nel,), dt), DeviceArray((nel,), dt)]
    else:
        bufs_in = [DeviceArray((nel,), dt)]
        bufs_out = bufs_in if inplace else [DeviceArray((nel,), dt)]
    # fill with random data by tiling one host block
    import ctypes
    from pyfft_amd import _native as N
    for b in bufs_in:
        nb = b.nbytes
        hb = host.view(numpy.uint8)[:min(nb, host.nbytes)]
        N.check(N.lib.mifft_memcpy_h2d(b.ptr, hb.ctypes.data, hb.nbytes, None))
        done = hb.nbytes
        while done < nb:
            n = min(done, nb - done)
            N.check(N.lib.mifft_memcpy_d2d(b.ptr + done, b.ptr, n, None))
            done += n
        N.check(N.lib.mifft_device_sync())
    plan.timed_execute(1, inplace, False, batch, bufs_in, bufs_out)  # warm-up
    # blocks of at least ~20 ms of back-to-back executes: a burst of a few milliseconds behind a synchronisation measures the
    # clock ramp, not the kernel (round 4: 2^20 x 128 0.37 in blocks of 5, 0.41 in blocks of 10 x 5 on the same kind of box)
    est = plan.timed_execute(iters, inplace, False, batch, bufs_in, bufs_out) / iters
    iters = int(min(200, max(iters, 20.0 / max(est, 1e-3))))
    best = 1e30
    for _ in range(3):
        ms = plan.timed_execute(iters, inplace, False, batch, bufs_in, bufs_out) / iters
        best = min(best, ms)
    alg = 2.0 * nel * itemsize
    flops = 5.0 * size * numpy.log2(size) * batch
    print("%-18s %-10s batch %-7d %s %s passes=%s  %.3f ms  %.1f GB/s alg (%.1f%% of 8TB/s)  %.0f GFLOPS" % (
        str(shape), dt.name, batch, "inpl" if inplace else "outp", plan.strategy(batch)[0], plan.pass_list(), best,
        alg / best / 1e6, alg / best / 1e6 / 80.0, flops / best / 1e6), flush=True)

if __name__ == "__main__":
    p = device_props()
    print(p.name.decode(), p.gcn_arch.decode(), "CUs", p.compute_units, flush=True)
    c64, c128, f32, f64 = numpy.complex64, numpy.complex128, numpy.float32, numpy.float64
    if len(sys.argv) > 4 and sys.argv[1] == "one":      # one SHAPE(1048576 or 1024x1024) DTYPE BATCH [inplace]
        cases = []
        run(tuple(int(t) for t in sys.argv[2].split("x")), numpy.dtype(sys.argv[3]).type, int(sys.argv[4]), inplace="inplace" in sys.argv[5:])
    elif len(sys.argv) > 1 and sys.argv[1] == "cfg":
        cases = [((1024, 1024), c64, 512), ((256, 256, 256), c128, 32), ((1 << 22,), c64, 128), ((1 << 16,), c64, 8192), ((1 << 18,), c64, 2048)]
    elif len(sys.argv) > 1 and sys.argv[1] == "f64":
        cases = [((1 << k,), c128, (1 << 27) >> k) for k in (14, 15, 16, 17, 18, 20, 22)] + [((1024, 1024), c128, 128), ((2048, 2048), c64, 64)]
    elif len(sys.argv) > 1 and sys.argv[1] == "3d":
        cases = [((128, 128, 128), c64, 128), ((32, 32, 128), c64, 2048), ((16, 16, 128), c64, 8192), ((64, 64, 64), c128, 512),
                 ((256, 64, 64), c64, 256), ((128, 128, 128), f32, 128), ((128, 64, 64), c128, 128)]
    elif len(sys.argv) > 1 and sys.argv[1] == "mid1d":
        cases = [((1 << k,), c64, (1 << 28) >> k) for k in (10, 11, 12, 13, 14)] + [((1 << k,), c128, (1 << 27) >> k) for k in (11, 12, 13)]
    elif len(sys.argv) > 1 and sys.argv[1] == "small":
        cases = [((16, 16), c64, 1 << 20), ((64, 64), c64, 1 << 16), ((16, 16, 16), c64, 1 << 16), ((8, 8, 64), c64, 1 << 16),
                 ((32, 16, 8), c64, 1 << 16), ((16, 16), c128, 1 << 19), ((16, 16), f32, 1 << 20), ((8, 8, 16), f64, 1 << 17),
                 ((4, 1024), c64, 1 << 16), ((2, 2), c64, 1 << 24), ((128, 64), c64, 1 << 15), ((128, 128), c64, 1 << 14),
                 ((16, 16, 64), c64, 1 << 14), ((16, 16, 32), c128, 1 << 14), ((64, 64), c128, 1 << 15)]
    elif len(sys.argv) > 1 and sys.argv[1] == "split":
        cases = [((1 << 20,), f32, 256), ((1 << 20,), c64, 256), ((1024, 1024), f32, 256), ((1 << 16,), f32, 4096), ((1 << 18,), f32, 1024),
                 ((1024,), f32, 1 << 16), ((256, 256), f32, 4096), ((128, 128, 128), f32, 128)]
    elif len(sys.argv) > 1 and sys.argv[1] == "nd":
        cases = [((1024, 1024), c64, 512), ((256, 256), c64, 8192), ((4096, 64), c64, 2048), ((128, 128, 128), c64, 256),
                 ((256, 256, 256), c128, 16), ((256, 256, 256), f64, 16), ((64, 64, 64), c128, 1024), ((1 << 20,), f32, 256),
                 ((1024, 1024), f32, 256)]
    elif len(sys.argv) > 1 and sys.argv[1] == "huge":
        cases = [((32, 32, 32), c64, 8192), ((256, 128), c64, 8192), ((128, 256), c64, 8192), ((32, 1024), c64, 8192), ((1024, 32), c64, 8192),
                 ((128, 128), c128, 8192), ((64, 256), c128, 8192), ((32, 32, 16), c128, 8192), ((128, 128, 128), c128, 64),
                 ((32, 32, 32), f32, 8192), ((128, 128), f64, 8192)]
    elif len(sys.argv) > 1 and sys.argv[1] == "ab":      # store-policy A/B set (MIFFT_STORE): single-pass rows / N-D, fused, pairs
        cases = [((1024,), c64, 1 << 17), ((4096,), c64, 1 << 15), ((16384,), c64, 1 << 13), ((32768,), c64, 1 << 12),
                 ((4096,), c128, 1 << 14), ((16,), c64, 1 << 23), ((16, 16, 16), c64, 1 << 15), ((128, 128), c64, 1 << 13),
                 ((128, 128), c128, 1 << 12), ((1 << 20,), c64, 512), ((1024, 1024), c64, 256), ((256, 256, 256), c128, 16),
                 ((1 << 22,), c64, 128), ((1 << 16,), c64, 8192)]
    elif len(sys.argv) > 1 and sys.argv[1] in ("1d1g", "1d8g"):      # the two-pass fp32 sizes at 1 GiB / 8 GiB per side (round 4)
        gib = 1 if sys.argv[1] == "1d1g" else 8
        cases = [((1 << k,), c64, (gib << 27) >> k) for k in (16, 17, 18, 19, 20, 21, 22)]
    elif len(sys.argv) > 1 and sys.argv[1] == "r4":                  # round-4 shapes: cubes, rectangles, fp64 2^21 at 1 GiB and 4 GiB
        cases = []
        for gib in (1, 4):
            cases += [((128, 128, 128), c64, gib * 64), ((128, 128, 128), c128, gib * 32), ((512, 1024), c64, gib * 256), ((1024, 512), c64, gib * 256),
                      ((1024, 2048), c64, gib * 64), ((2048, 1024), c64, gib * 64), ((2048, 512), c64, gib * 128), ((512, 2048), c64, gib * 128),
                      ((1 << 21,), c128, gib * 32), ((1 << 22,), c128, gib * 16), ((1 << 20,), c128, gib * 64)]
    elif len(sys.argv) > 1 and sys.argv[1] == "r4b":                 # second batch of round 4, 2 GiB per side
        def nb(shape, dt):
            d = numpy.dtype(dt)
            return (2 << 30) // (int(numpy.prod(shape)) * d.itemsize * (2 if d.kind == "f" else 1))
        shapes = [((256, 256), c64), ((256, 512), c64), ((512, 256), c64), ((256, 1024), c64), ((1024, 256), c64), ((256, 256), c128), ((512, 256), c128)] + \
            [((a, b, c), dt) for dt in (c64, c128) for a in (64, 128) for b in (64, 128) for c in (64, 128)] + \
            [((64, 64, 64), f32), ((64, 128, 128), f32), ((128, 128, 128), f32), ((64, 64, 64), f64), ((128, 128, 128), f64), ((128, 128, 64), f64)] + \
            [((1 << k,), f32) for k in (16, 17, 18, 19, 20)] + [((1 << 20,), f64), ((1024, 1024), f32)]
        cases = [(s, d, nb(s, d)) for s, d in shapes]
    elif len(sys.argv) > 1 and sys.argv[1] == "tail":                # shapes that still run two or three launches per cache-sized chunk (round 4 survey)
        cases = [((256, 256), c64, 4096), ((256, 256), c128, 2048), ((512, 256), c64, 2048), ((64, 64, 64), c64, 1024), ((64, 64, 64), c128, 512),
                 ((128, 128, 64), c64, 256), ((64, 128, 128), c64, 256), ((256, 128, 128), c64, 64), ((128, 128, 64), c128, 128),
                 ((256, 256, 128), c64, 32), ((256, 256, 256), c64, 16), ((1 << 15,), c128, 4096), ((256, 4096), c64, 256), ((4096, 256), c64, 256),
                 ((4096, 4096), c64, 16), ((1 << 20,), f32, 256), ((1 << 18,), f32, 1024), ((1024, 1024), f32, 256), ((128, 128, 128), f32, 128),
                 ((1 << 20,), f64, 128), ((32, 32, 2048), c64, 128), ((2048, 32, 32), c64, 128), ((1 << 23,), c64, 32), ((1 << 24,), c64, 16),
                 ((512, 512, 512), c64, 2), ((2048, 2048), c128, 32)]
    elif len(sys.argv) > 1 and sys.argv[1] == "r5":                  # round 5: pass-pair chains, several work-groups per transform (1 GiB per side)
        def nb(shape, dt, gib=1.0):
            d = numpy.dtype(dt)
            return max(1, int(gib * (1 << 30)) // (int(numpy.prod(shape)) * d.itemsize))
        shapes = [((4096, 256), c64), ((4096, 1024), c64), ((4096, 4096), c64), ((32, 32, 2048), c64), ((16, 16, 2048), c64), ((4096, 512), c128), ((32, 32, 1024), c128),
                  ((16, 16, 128), c64), ((128, 256), c64), ((256, 128), c64), ((512, 64), c64), ((8, 64, 64), c64), ((32, 1024), c64), ((1024, 32), c64), ((32, 32, 32), c64),
                  ((64, 256), c128), ((256, 64), c128), ((16, 32, 32), c128), ((16, 16, 64), c128), ((32, 512), c128), ((512, 32), c128), ((128, 128), c128),
                  ((256, 256), c64), ((512, 128), c64), ((1024, 64), c64), ((64, 1024), c64), ((16, 64, 64), c64)]
        cases = [(s, d, nb(s, d)) for s, d in shapes] + [(s, d, nb(s, d, 0.03125)) for s, d in shapes[7:22]] + \
            [((128, 128), c64, 256), ((128, 128), c128, 128), ((256, 64), c64, 256), ((16, 32, 32), c64, 256), ((128, 64), c128, 256)]
    elif len(sys.argv) > 1 and sys.argv[1] == "1d":
        cases = [((1 << k,), c64, max(64, (1 << 31) >> (k + 3))) for k in (13, 14, 15, 16, 17, 18, 19, 20, 21, 22)]
    else:
        cases = [
            ((1024,), c64, 1 << 16), ((4096,), c64, 1 << 14), ((256,), c64, 1 << 18), ((16,), c64, 1 << 22),
            ((1 << 20,), c64, 256), ((1 << 20,), f32, 64),
            ((1 << 16,), c64, 1024), ((8192,), c64, 8192),
            ((1024, 1024), c64, 256),
            ((256, 256, 256), c128, 4), ((256, 256, 256), f64, 4), ((128, 128, 128), c64, 32),
            ((1 << 22,), c64, 32),
        ]
    for shape, dt, batch in cases:
        try:
            run(shape, dt, batch)
        except Exception as e:
            print("FAILED", shape, dt, batch, repr(e), flush=True)

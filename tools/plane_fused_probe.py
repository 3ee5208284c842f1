#!/usr/bin/env python3
"""Round 6: 3-D transforms bigger than a pipeline chunk, (y, x) planes on the persistent 2-D kernels + plain z launches (strategy
"fused2z") against the route of rounds 1-5 (PYFFT_AMD_NO_PLANE_FUSED=1: three plain launches, or the leading passes slab by slab through the
pipelined launcher), and with the far-apart rows of the z pass on 16-column tiles (MIFFT_NARROW_TILES=3), out of place on random data, best of three blocks of four executes between two HIP events.

    python3 tools/plane_fused_probe.py
"""
import os
import sys

import numpy

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfft_amd import _native as N                      # noqa: E402
from pyfft_amd.hip import DeviceArray, Plan, device_props      # noqa: E402

CASES = [((512, 512, 512), "complex64", 2), ((512, 512, 512), "complex64", 4), ((256, 512, 512), "complex64", 8),
         ((64, 512, 512), "complex64", 16), ((64, 1024, 1024), "complex64", 4), ((16, 1024, 1024), "complex64", 16),
         ((128, 1024, 512), "complex64", 8), ((64, 2048, 2048), "complex64", 2), ((128, 512, 256), "complex64", 16),
         ((64, 1024, 256), "complex64", 16), ((512, 512, 512), "complex128", 2), ((64, 512, 512), "complex128", 8),
         ((64, 512, 256), "complex128", 16), ((16, 1024, 512), "complex128", 16), ((64, 1024, 1024), "complex128", 2),
         ((64, 1024, 1024), "float64", 2), ((8, 1024, 1024), "float64", 16)]


def fill(buf, seed):
    rng = numpy.random.default_rng(seed)
    host = rng.standard_normal(1 << 22).astype(numpy.float32).view(numpy.uint8)
    n0 = min(buf.nbytes, host.nbytes)
    N.check(N.lib.mifft_memcpy_h2d(buf.ptr, host.ctypes.data, n0, None))
    done = n0
    while done < buf.nbytes:
        n = min(done, buf.nbytes - done)
        N.check(N.lib.mifft_memcpy_d2d(buf.ptr + done, buf.ptr, n, None))
        done += n
    N.check(N.lib.mifft_device_sync())


def run(shape, dtype, batch, env, a, b):
    for k in ("PYFFT_AMD_NO_PLANE_FUSED", "MIFFT_NARROW_TILES"):
        os.environ.pop(k, None)
    os.environ.update(env)
    N.check(N.lib.mifft_debug_set(N.DEBUG_NARROW_TILES, int(env.get("MIFFT_NARROW_TILES", 0))), "debug_set")
    dt = numpy.dtype(dtype)
    split = dt.kind == "f"
    n = int(numpy.prod(shape)) * batch
    plan = Plan(shape, dtype=dtype)
    bi, bo = (a + [None])[:2], (b + [None])[:2]
    plan.timed_execute(2, False, False, batch, bi, bo)
    best = min(plan.timed_execute(4, False, False, batch, bi, bo) / 4 for _ in range(3))
    N.check(N.lib.mifft_debug_set(N.DEBUG_NARROW_TILES, 0), "debug_set")
    return 2.0 * n * dt.itemsize * (2 if split else 1) / (best * 1e-3) / 8e12, plan.strategy(batch)


def main():
    p = device_props()
    print(p.name.decode(), p.gcn_arch.decode(), "CUs", p.compute_units, flush=True)
    print("fraction of the 8 TB/s roofline, out of place, random data")
    for shape, dtype, batch in CASES:
        dt = numpy.dtype(dtype)
        n = int(numpy.prod(shape)) * batch
        a = [DeviceArray((n,), dt) for _ in range(2 if dt.kind == "f" else 1)]
        b = [DeviceArray((n,), dt) for _ in range(2 if dt.kind == "f" else 1)]
        for i, x in enumerate(a):
            fill(x, 7 + i)
        f1, s1 = run(shape, dtype, batch, {}, a, b)
        f2, _ = run(shape, dtype, batch, {"MIFFT_NARROW_TILES": "3"}, a, b)
        f0, s0 = run(shape, dtype, batch, {"PYFFT_AMD_NO_PLANE_FUSED": "1"}, a, b)
        print("%-18s %-10s x %-3d  planes on the persistent kernel %.3f %s  (z pass on 16-column tiles: %.3f)   before %.3f %s" % (
            shape, dtype, batch, f1, s1, f2, f0, s0), flush=True)
        del a, b


if __name__ == "__main__":
    main()

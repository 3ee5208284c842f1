#!/usr/bin/env python3
"""Round 6, review item 4: the reference's published N-D shapes in its split-complex layout (float32 / float64 planes) against their
interleaved twins, at 1 GiB and at the reference's 32 MiB per side: the dense kernel on 16-byte plane accesses (csrc/fft_nd2p.hpp) and
what ran before it (MIFFT_DEBUG_ALT_ROWS = 7: the tiled fixed-shape kernel with one tile per parent / the run-time-shaped kernel).

    python3 tools/planes_probe.py
"""
import os
import sys

import numpy

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfft_amd import _native as N                      # noqa: E402
from pyfft_amd.hip import DeviceArray, Plan, device_props      # noqa: E402

SHAPES = [(16, 16), (32, 32), (64, 64), (128, 128), (16, 16, 16), (8, 8, 64), (16, 16, 128), (32, 32, 32)]
# float32 shapes of 32768 points: out of place on two half-size work-groups per transform (csrc/fft_nd2zp.hpp); `python3 tools/planes_probe.py halves`
HALVES = [(16, 16, 128), (128, 256), (256, 128), (512, 64), (64, 512), (16, 2048), (2048, 16), (8, 64, 64), (64, 8, 64), (16, 128, 16), (128, 16, 16),
          (16, 32, 64), (32, 16, 64), (16, 64, 32), (64, 16, 32), (32, 64, 16), (256, 256), (512, 128), (1024, 64), (64, 1024), (16, 64, 64)]


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


def measure(shape, dtype, side_bytes, alt):
    dt = numpy.dtype(dtype)
    split = dt.kind == "f"
    size = int(numpy.prod(shape))
    csz = dt.itemsize * (2 if split else 1)
    batch = max(1, side_bytes // (size * csz))
    nel = size * batch
    N.check(N.lib.mifft_debug_set(N.DEBUG_ALT_ROWS, alt), "debug_set")
    try:
        plan = Plan(shape, dtype=dtype)
        ins = [DeviceArray((nel,), dt) for _ in range(2 if split else 1)]
        outs = [DeviceArray((nel,), dt) for _ in range(2 if split else 1)]
        for i, b in enumerate(ins):
            fill(b, 40 + i)
        bi = (ins + [None])[:2] if split else [ins[0], None]
        bo = (outs + [None])[:2] if split else [outs[0], None]
        plan.timed_execute(2, False, False, batch, bi, bo)
        est = plan.timed_execute(5, False, False, batch, bi, bo) / 5
        it = int(min(400, max(5, 20.0 / max(est, 1e-3))))
        best = min(plan.timed_execute(it, False, False, batch, bi, bo) / it for _ in range(3))
        launches = len(plan.pass_list(inplace=False))
    finally:
        N.check(N.lib.mifft_debug_set(N.DEBUG_ALT_ROWS, 0), "debug_set")
    return 2.0 * nel * csz / (best * 1e-3) / 8e12, launches


def main():
    p = device_props()
    print(p.name.decode(), p.gcn_arch.decode(), "CUs", p.compute_units, flush=True)
    print("fraction of the 8 TB/s roofline, out of place; planes16 = csrc/fft_nd2p.hpp, before = MIFFT_DEBUG_ALT_ROWS 7; (n) = launches per execute")
    for side, name in ((1 << 30, "1 GiB per side"), (32 << 20, "32 MiB per side (the reference's protocol)")):
        print("## " + name)
        for cd, rd in ((numpy.complex64, numpy.float32), (numpy.complex128, numpy.float64)):
            for shape in (HALVES if "halves" in sys.argv[1:] else SHAPES):
                if rd == numpy.float64 and (int(numpy.prod(shape)) > 16384 or "halves" in sys.argv[1:]):
                    continue
                twin, lt = measure(shape, cd, side, 0)
                new, ln = measure(shape, rd, side, 0)
                old, lo = measure(shape, rd, side, 7)
                print("%-14s %-8s interleaved %.3f (%d)   planes16 %.3f (%d) = %.2f of the twin   before %.3f (%d) = %.2f" % (
                    str(shape), numpy.dtype(rd).name, twin, lt, new, ln, new / twin, old, lo, old / twin), flush=True)


if __name__ == "__main__":
    main()

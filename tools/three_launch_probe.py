#!/usr/bin/env python3
"""Round 6, review item 6: would fusing two of the three launches of the longest 1-D sizes pay?  Measured ONCE, by composition.

N = 2^23 runs three strided passes today (256 x 256 x 128).  The proposal: two 1024-point passes in ONE persistent launch + one plain
radix-8 pass.  Whatever order the three factors take, a tile of the second fused pass needs output of EVERY tile of the first one (the
8 interleaved 2^20-point sub-problems share every 16-column tile), so the hand-over unit is the whole 64 MiB transform: a ring of three
slots, the form that lost on fp64 2^22 (0.243-0.259 against 0.269, profiles/r04_e_fp64_long_fused.log).  This probe therefore measures an
UPPER BOUND of the proposal from launches that exist, moving exactly the proposal's bytes:

    (a) the persistent two-pass launch of 8 B transforms of 2^20 points (the two 1024-point passes with an 8 MiB hand-over unit -- better
        than the proposal could have), then
    (b) ONE plain strided radix-8 (2^24: radix-16) pass over the same B x 2^23 points in place: col(L = 8, M = 1, S = 2^20) through the C ABI,

timed back to back between two HIP events, against the plan's own three launches on the same buffers.  (The composite is no transform of
2^23 points -- its inter-pass twiddles are those of 2^20 -- only its data movement is the proposal's.)

The same arithmetic for 512^3 (1 GiB per transform in complex64: x rows + y columns + z columns, three launches, no pass pair for a
512-point y axis): x-row launch + ONE persistent launch over (y, z).  Here the hand-over unit IS small -- 16 x-columns of every (y, z)
plane, 32 MiB -- so the ring rule carries over; the bound is composed of (a) the plan of 512-point rows over the same points and (b)
the persistent 2-D launch of (512, 512) transforms over the same bytes (16-byte ... 4 KiB runs instead of 128-byte segments 4 KiB apart:
an upper bound again).

    python3 tools/three_launch_probe.py [gib_per_side]
"""
import ctypes
import os
import sys

import numpy

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfft_amd import _native as N                      # noqa: E402
from pyfft_amd.hip import DeviceArray, Event, Plan, device_props      # noqa: E402
from pyfft_amd.plan import _twiddle_table               # noqa: E402


def fill(buf):
    rng = numpy.random.default_rng(3)
    host = rng.standard_normal(1 << 23).astype(numpy.float32).view(numpy.uint8)
    N.check(N.lib.mifft_memcpy_h2d(buf.ptr, host.ctypes.data, min(buf.nbytes, host.nbytes), None))
    done = min(buf.nbytes, host.nbytes)
    while done < buf.nbytes:
        n = min(done, buf.nbytes - done)
        N.check(N.lib.mifft_memcpy_d2d(buf.ptr + done, buf.ptr, n, None))
        done += n
    N.check(N.lib.mifft_device_sync())


def timed(stream, fn, reps):
    fn()
    e0, e1 = Event(), Event()
    best = 1e30
    for _ in range(3):
        e0.record(stream)
        for _ in range(reps):
            fn()
        e1.record(stream)
        e1.synchronize()
        best = min(best, e1.time_since(e0) / reps)
    return best


def main():
    gib = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
    p = device_props()
    print(p.name.decode(), p.gcn_arch.decode(), "CUs", p.compute_units, flush=True)
    for logn, radix in ((23, 8), (24, 16)):
        n = 1 << logn
        batch = max(4, int(gib * (1 << 30)) // (n * 8))
        a = DeviceArray((n * batch,), numpy.complex64)
        b = DeviceArray((n * batch,), numpy.complex64)
        fill(a)
        plan3 = Plan(n, dtype=numpy.complex64, wait_for_finish=False)
        stream = None

        def three():
            plan3.execute(a, b, batch=batch)
        three()
        stream = plan3._context.getQueue()
        plan3.finish()
        ms3 = timed(stream, three, 4)
        plan3.finish()
        # (a) the persistent two-pass launch on radix * batch transforms of 2^20 points, on the same stream
        sub = batch * radix
        plan2 = Plan(1 << 20, dtype=numpy.complex64, stream=stream)
        # (b) the plain radix pass: col(L = radix, M = 1, S = 2^20), in place on b
        tw = DeviceArray((radix,), numpy.complex64).set(_twiddle_table(radix, radix, 1, numpy.dtype(numpy.complex64)))
        d = N.MifftPass()
        d.kind, d.precision, d.layout, d.inverse, d.L, d.variant = N.PASS_COL, N.F32, N.INTERLEAVED, 0, radix, 0
        d.M, d.S, d.outer, d.outer_stride_in, d.outer_stride_out, d.scale = 1, 1 << 20, batch, n, n, 1.0
        d.tw_L = tw.ptr
        d.flags = N.FLAG_STREAM_DST

        def fused_two():
            plan2.execute(a, b, batch=sub)

        def radix_pass():
            N.check(N.lib.mifft_launch_pass(ctypes.byref(d), b.ptr, None, b.ptr, None, stream.handle if hasattr(stream, "handle") else stream), "launch_pass")

        def composite():
            fused_two()
            radix_pass()
        ms_a = timed(stream, fused_two, 4)
        ms_b = timed(stream, radix_pass, 8)
        ms_c = timed(stream, composite, 4)
        plan2.finish()
        alg = 2.0 * n * batch * 8
        frac = lambda ms: alg / (ms * 1e-3) / 8e12      # noqa: E731
        print("N = 2^%d x %d (%.1f GiB per side)  plan: %s %s" % (logn, batch, n * batch * 8 / 2.0 ** 30, plan3.strategy(batch)[0], plan3.pass_list()))
        print("   the plan's three launches            %8.3f ms  %.3f of the roofline" % (ms3, frac(ms3)))
        print("   (a) persistent launch, 2^20 x %-6d %8.3f ms  %.3f   [%s]" % (sub, ms_a, frac(ms_a), plan2.strategy(sub)[0]))
        print("   (b) plain radix-%-2d pass, in place     %8.3f ms  %.3f" % (radix, ms_b, frac(ms_b)))
        print("   (a) + (b) back to back               %8.3f ms  %.3f   (1 / (1/a + 1/b) = %.3f)" % (ms_c, frac(ms_c), 1.0 / (1.0 / frac(ms_a) + 1.0 / frac(ms_b))), flush=True)
        del a, b


def cube512(batch=2):
    n = 512
    pts = n ** 3 * batch
    a = DeviceArray((pts,), numpy.complex64)
    b = DeviceArray((pts,), numpy.complex64)
    fill(a)
    plan3 = Plan((n, n, n), dtype=numpy.complex64, wait_for_finish=False)
    plan3.execute(a, b, batch=batch)
    stream = plan3._context.getQueue()
    plan3.finish()
    ms3 = timed(stream, lambda: plan3.execute(a, b, batch=batch), 4)
    plan3.finish()
    rows = Plan(n, dtype=numpy.complex64, stream=stream)
    planes = Plan((n, n), dtype=numpy.complex64, stream=stream)
    ms_a = timed(stream, lambda: rows.execute(a, b, batch=pts // n), 4)
    ms_b = timed(stream, lambda: planes.execute(b, batch=pts // (n * n)), 4)

    def composite():
        rows.execute(a, b, batch=pts // n)
        planes.execute(b, batch=pts // (n * n))
    ms_c = timed(stream, composite, 4)
    rows.finish()
    # the other cut, and no bound but the real thing: the persistent 2-D launch over the batch * 512 (y, x) planes, out of place, then the
    # plan's own strided z pass in place -- col(L = 512, M = 1, S = 512 * 512) through the C ABI: together a 3-D transform of the data
    tw = DeviceArray((n,), numpy.complex64).set(_twiddle_table(n, n, 1, numpy.dtype(numpy.complex64)))
    d = N.MifftPass()
    d.kind, d.precision, d.layout, d.inverse, d.L, d.variant = N.PASS_COL, N.F32, N.INTERLEAVED, 0, n, 0
    d.M, d.S, d.outer, d.outer_stride_in, d.outer_stride_out, d.scale = 1, n * n, batch, n ** 3, n ** 3, 1.0
    d.tw_L = tw.ptr
    d.flags = N.FLAG_STREAM_DST
    sh = stream.handle if hasattr(stream, "handle") else stream

    def z_pass():
        N.check(N.lib.mifft_launch_pass(ctypes.byref(d), b.ptr, None, b.ptr, None, sh), "launch_pass")

    def planes_then_z():
        planes.execute(a, b, batch=pts // (n * n))
        z_pass()
    ms_p = timed(stream, lambda: planes.execute(a, b, batch=pts // (n * n)), 4)
    ms_z = timed(stream, z_pass, 4)
    ms_pz = timed(stream, planes_then_z, 4)
    planes.finish()
    # values: the composite is the transform (checked on the first transform against the plan's result)
    plan3.execute(a, b, batch=batch)
    plan3.finish()
    ref = b.get()[:n ** 3].copy()
    planes_then_z()
    planes.finish()
    got = b.get()[:n ** 3]
    err = float(numpy.abs(got - ref).max() / numpy.abs(ref).max())
    alg = 2.0 * pts * 8
    frac = lambda ms: alg / (ms * 1e-3) / 8e12      # noqa: E731
    print("(512, 512, 512) x %d (%.1f GiB per side)  plan: %s %s" % (batch, pts * 8 / 2.0 ** 30, plan3.strategy(batch)[0], plan3.pass_list()))
    print("   the plan's three launches            %8.3f ms  %.3f of the roofline" % (ms3, frac(ms3)))
    print("   (a) 512-point rows, out of place     %8.3f ms  %.3f   [%s]" % (ms_a, frac(ms_a), rows.strategy(pts // n)[0]))
    print("   (b) persistent (512, 512), in place  %8.3f ms  %.3f   [%s]" % (ms_b, frac(ms_b), planes.strategy(pts // (n * n))[0]))
    print("   (a) + (b) back to back               %8.3f ms  %.3f   (1 / (1/a + 1/b) = %.3f)" % (ms_c, frac(ms_c), 1.0 / (1.0 / frac(ms_a) + 1.0 / frac(ms_b))), flush=True)
    print("   (c) persistent (512, 512), out of place %5.3f ms  %.3f" % (ms_p, frac(ms_p)))
    print("   (d) the plan's z pass, in place      %8.3f ms  %.3f" % (ms_z, frac(ms_z)))
    print("   (c) + (d) = the transform            %8.3f ms  %.3f   max |difference| to the plan's result %.2e of max |value|" % (ms_pz, frac(ms_pz), err), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "cube":
        cube512(2)
        cube512(4)
    else:
        main()
        cube512(2)
        cube512(4)

"""Round 5, review item 2: anti-phase XCD groups for the XCD-resident single-crossing kernel (csrc/fft_xcd2.hpp).

The XCDs with an odd HW_REG_XCC_ID start `delay` microseconds late (flags bits 24..30 of mifft_launch_xcd2), so that about half of the
XCDs are in their HBM burst while the other half exchange through L2.  For every delay: device time of the launch (bit-identical
results, checked against the chain), and from the per-work-group time stamps of three transforms along the run the burst interval
[stamp 11, stamp 12) of every XCD -- how many XCDs burst at the same time, how long a burst takes, and whether the offset survives
(drift).      python3 tools/xcd2_antiphase.py [batch ...]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
from pyfft_amd import _native as N

n = 1 << 20
BATCHES = [int(a) for a in sys.argv[1:]] or [512, 2048]
DELAYS = (0, 8, 16, 24, 32, 40)
os.environ["PYFFT_AMD_STRATEGY"] = "xcd"
from pyfft_amd.hip import Plan, DeviceArray

rng = numpy.random.default_rng(7)
blk = 8
host = (rng.standard_normal((blk, n)) + 1j * rng.standard_normal((blk, n))).astype(numpy.complex64)


def plan_for(strategy, flags=None):
    os.environ["PYFFT_AMD_STRATEGY"] = strategy
    if flags is None:
        os.environ.pop("PYFFT_AMD_XCD2_FLAGS", None)
    else:
        os.environ["PYFFT_AMD_XCD2_FLAGS"] = str(flags)
    return Plan((n,), dtype=numpy.complex64, wait_for_finish=True)


def frac(ms, B):
    return 16.0 * n * B / (ms * 1e-3) / 8e12


for B in BATCHES:
    a = DeviceArray((B * n,), numpy.complex64)
    for i in range(0, B, blk):
        N.check(N.lib.mifft_memcpy_h2d(a.ptr + i * n * 8, host.ctypes.data, min(blk, B - i) * n * 8, None))
    ref = DeviceArray((B * n,), numpy.complex64)
    out = DeviceArray((B * n,), numpy.complex64)
    p = plan_for("chain")
    p.timed_execute(1, False, False, B, [a, None], [ref, None])
    h_ref = ref.get().view(numpy.uint32)
    p = plan_for("fused")
    p.timed_execute(1, False, False, B, [a, None], [out, None])
    ms = min(p.timed_execute(5, False, False, B, [a, None], [out, None]) / 5 for _ in range(3))
    print("N = 2^20 complex64, batch %d (%d transforms per XCD).  fused2 (two crossings): %.3f ms  %.3f of 8 TB/s" % (B, B // 8, ms, frac(ms, B)), flush=True)
    per_xcd = B // 8
    for delay in DELAYS:
        flags = 1 | (delay << 24)
        p = plan_for("xcd", flags)
        assert p.strategy(B)[0] == "xcd2"
        N.check(N.lib.mifft_memset(out.ptr, 0, out.nbytes, None))
        p.timed_execute(1, False, False, B, [a, None], [out, None])
        same = numpy.array_equal(h_ref, out.get().view(numpy.uint32))
        ms = min(p.timed_execute(5, False, False, B, [a, None], [out, None]) / 5 for _ in range(3))
        print("odd XCDs start %2d us late: %.3f ms  %6.2f us per transform and XCD  %.3f of 8 TB/s   bit-identical to the chain: %s" % (
            delay, ms, ms * 1e3 / per_xcd, frac(ms, B), same), flush=True)
        if delay not in (0, 24):
            continue
        for it in (4, per_xcd // 2, per_xcd - 4):
            pt = plan_for("xcd", flags | N.XCD2_TRACE | (it << 8))
            pt.timed_execute(1, False, False, B, [a, None], [out, None])
            raw = numpy.zeros(512 * 32, numpy.uint64)
            N.check(N.lib.mifft_memcpy_d2h(raw.ctypes.data, pt._context.pointer_of(pt._counters) + N.XCD2_CONTROL_BYTES, raw.nbytes, None))
            t = raw.reshape(8, 64, 32).astype(numpy.float64) / 100.0      # us; [xcd][work-group][stamp]
            t0 = t[:, :, 0].min()
            line = []
            ivals = []
            for x in range(8):
                start = t[x, :, 0].mean() - t0
                b0 = t[x, :, 11].mean() - t0
                b1 = t[x, :, 12].mean() - t0
                ivals.append((b0, b1))
                line.append("x%d start %6.1f burst [%6.1f, %6.1f) %5.1f us, whole %5.1f" % (x, start, b0, b1, b1 - b0, t[x, :, 12].mean() - t[x, :, 0].mean()))
            # how many XCDs are bursting at one time: the XCDs are at different transform indices at one wall time, so fold the
            # intervals onto one period (the mean whole-transform time) and count around the circle
            period = numpy.mean([t[x, :, 12].mean() - t[x, :, 0].mean() for x in range(8)])
            grid = numpy.linspace(0.0, period, 400, endpoint=False)
            conc = numpy.array([sum(1 for b, e in ivals if ((g - b) % period) < (e - b)) for g in grid])
            print("   per-XCD transform %4d: mean burst %.1f us, mean whole transform %.1f us, bursts at one time (folded onto one period): mean %.2f min %d max %d" % (
                it, numpy.mean([e - b for b, e in ivals]), numpy.mean([t[x, :, 12].mean() - t[x, :, 0].mean() for x in range(8)]), conc.mean(), conc.min(), conc.max()))
            for s in line:
                print("      " + s)
    del a, ref, out

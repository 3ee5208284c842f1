"""Development probe for the XCD-cooperative strategy (fft_xcd2.hpp): bit-identity against the chain strategy and
device time against fused2 / pipelined, N = 2^20 fp32.  Usage: python3 tools/xcd2_probe.py [batch] [reps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
from pyfft_amd.hip import Plan, DeviceArray
from pyfft_amd import _native as N

n = 1 << 20
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rng = numpy.random.default_rng(7)
blk = 8
host = (rng.standard_normal((blk, n)) + 1j * rng.standard_normal((blk, n))).astype(numpy.complex64)
a = DeviceArray((B * n,), numpy.complex64)
for i in range(0, B, blk):
    N.check(N.lib.mifft_memcpy_h2d(a.ptr + i * n * 8, host.ctypes.data, min(blk, B - i) * n * 8, None))
ref = DeviceArray((B * n,), numpy.complex64)
out = DeviceArray((B * n,), numpy.complex64)


def run(strategy, flags=None, inverse=False, dst=None, inplace=False):
    os.environ["PYFFT_AMD_STRATEGY"] = strategy
    if flags is None:
        os.environ.pop("PYFFT_AMD_XCD2_FLAGS", None)
    else:
        os.environ["PYFFT_AMD_XCD2_FLAGS"] = str(flags)
    plan = Plan((n,), dtype=numpy.complex64, wait_for_finish=True)
    st = plan.strategy(B)
    src = [a, None]
    d = [dst if dst is not None else out, None]
    if inplace:
        d = src
    ms = plan.timed_execute(1, inplace, inverse, B, src, d)            # warm-up
    best = min(plan.timed_execute(reps, inplace, inverse, B, src, d) / reps for _ in range(3))
    frac = 16.0 * n * B / (best * 1e-3) / 8e12
    print("%-10s flags=%s %s: strategy %s  %.3f ms  %.1f%% of 8 TB/s" % (strategy, flags, "inv" if inverse else "fwd", st[0], best, 100 * frac), flush=True)
    return plan


run("chain", dst=ref)
h_ref = ref.get()
for flags in (1, 0):
    out_h0 = None
    p = run("xcd", flags)
    h = out.get()
    same = numpy.array_equal(h_ref.view(numpy.uint32), h.view(numpy.uint32))
    want = numpy.fft.fft(host[1].astype(numpy.complex128))
    got = h[n:2 * n].astype(numpy.complex128)
    print("   bit-identical to chain: %s   max rel err vs numpy (item 1): %.3e   last item equal: %s" % (
        same, numpy.abs(got - want).max() / numpy.abs(want).max(),
        numpy.array_equal(h_ref[-n:].view(numpy.uint32), h[-n:].view(numpy.uint32))), flush=True)
    N.check(N.lib.mifft_memset(out.ptr, 0, out.nbytes, None))
run("fused")
run("pipelined")
# inverse, bit-identity again
run("chain", inverse=True, dst=ref)
h_ref = ref.get()
run("xcd", 1, inverse=True)
print("   inverse bit-identical: %s" % numpy.array_equal(h_ref.view(numpy.uint32), out.get().view(numpy.uint32)))

# ---- phase trace of one transform (development flag MIFFT_XCD2_TRACE): mean over the 512 work-groups, microseconds
def trace(it):
    p = run("xcd", 1 | N.XCD2_TRACE | (it << 8))
    raw = numpy.zeros(512 * 32, numpy.uint64)
    N.check(N.lib.mifft_memcpy_d2h(raw.ctypes.data, p._context.pointer_of(p._counters) + N.XCD2_CONTROL_BYTES, raw.nbytes, None))
    t = raw.reshape(512, 32).astype(numpy.float64) / 100.0      # us
    t = t[t[:, 0] > 0]
    names = ["loads + stage 1,2", "(stamp)", "compute0 stores0 compute1 A0", "loads0 B0", "stores1 compute2 A1", "loads1 st1(y0) B1",
             "stores2 compute3 A2", "loads2 st1(y1) B2", "stores3 A3", "loads3 st1(y2) B3", "st1(y3) + stage 2",
             "final rounds (HBM stores + prefetch)"]
    d = numpy.diff(t[:, :13], axis=1)
    print("trace of per-XCD transform %d over %d work-groups: total %.2f us (min %.2f max %.2f)" % (
        it, len(t), (t[:, 12] - t[:, 0]).mean(), (t[:, 12] - t[:, 0]).min(), (t[:, 12] - t[:, 0]).max()))
    for i, nm in enumerate(names):
        print("   %-40s mean %6.2f  min %6.2f  max %6.2f" % (nm, d[:, i].mean(), d[:, i].min(), d[:, i].max()))
    # spread of the XCDs' phases: start time of this transform per XCD (first 8 work-groups' stamp 0 differ by XCD)
    print("   start-time spread over work-groups: %.2f us" % (t[:, 0].max() - t[:, 0].min()))


trace(20)
trace(40)

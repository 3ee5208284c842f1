"""XCD-local fused form (development strategy `fusedx`) against the plan's own choice: parity, time, and (under rocprofv3 --pmc)
traffic.  python3 tools/fusedx_probe.py [sweep|one LOG2N LAG RING WT]"""
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, numpy
sys.path.insert(0, %r)
from pyfft_amd.hip import Plan, DeviceArray, Event
from pyfft_amd import _native as N
n = 1 << int(sys.argv[1]); batch = (1 << 29) // n
a = DeviceArray((n * batch,), numpy.complex64); b = DeviceArray((n * batch,), numpy.complex64)
rng = numpy.random.default_rng(5)
blk = (rng.standard_normal((8, n)) + 1j * rng.standard_normal((8, n))).astype(numpy.complex64)
for i in range(0, batch, 8):
    N.check(N.lib.mifft_memcpy_h2d(a.ptr + i * n * 8, blk.ctypes.data, blk.nbytes, None))
plan = Plan(n, dtype=numpy.complex64, wait_for_finish=True)
plan.execute(a, b, batch=batch)
out = numpy.empty(n, numpy.complex64); worst = 0.0
for item in (0, 1, 7, 8, 9, batch // 2 + 3, batch - 1):
    N.check(N.lib.mifft_memcpy_d2h(out.ctypes.data, b.ptr + item * n * 8, n * 8, None))
    ref = numpy.fft.fft(blk[item %% 8].astype(numpy.complex128))
    worst = max(worst, numpy.abs(out - ref).sum() / numpy.abs(ref).sum())
st = plan._context.getQueue(); best = 1e9
for _ in range(3):
    e0 = Event().record(st)
    for _ in range(5): plan.execute(a, b, batch=batch, wait_for_finish=False)
    e1 = Event().record(st); e1.synchronize(); best = min(best, e1.time_since(e0) / 5)
plan.finish()
print("2^%%s x %%-5d %%-34s %%8.3f ms  %%.3f of roofline  err %%.1e" %% (sys.argv[1], batch, str(plan.strategy(batch)[:5]), best, 16.0 * n * batch / (best * 1e-3) / 8e12, worst))
''' % ROOT


def run(log2n, env):
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, "-c", CHILD, str(log2n)], env=e, capture_output=True, text=True)
    print(" ".join("%s=%s" % kv for kv in env.items()).ljust(60), r.stdout.strip() or r.stderr.strip()[-400:], flush=True)


if __name__ != "__main__":
    pass
elif len(sys.argv) > 1 and sys.argv[1] == "one":
    run(int(sys.argv[2]), {"PYFFT_AMD_STRATEGY": "fusedx", "PYFFT_AMD_FUSEDX": "%s,%s,%s" % tuple(sys.argv[3:6])})
else:
    for log2n in (16, 17, 18, 19, 20):
        run(log2n, {})
        for lag, ring in ((1, 2), (1, 3), (2, 3), (2, 4), (3, 6), (4, 8)):
            for wt in (0, 1):
                run(log2n, {"PYFFT_AMD_STRATEGY": "fusedx", "PYFFT_AMD_FUSEDX": "%d,%d,%d" % (lag, ring, wt)})

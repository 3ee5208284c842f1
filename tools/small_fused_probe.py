"""Small batches of the shapes that have a fused two-pass kernel: one launch per pass (chain) against the persistent launch with
one ring slot per transform (PYFFT_AMD_SMALL_FUSED = lag divisor).  python3 tools/small_fused_probe.py"""
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, numpy
sys.path.insert(0, %r)
from pyfft_amd.hip import Plan, DeviceArray, Event
shape = tuple(int(t) for t in sys.argv[1].split("x")); dtype = numpy.dtype(sys.argv[2]); batch = int(sys.argv[3])
size = int(numpy.prod(shape))
rng = numpy.random.default_rng(5)
data = (rng.standard_normal(size * batch) + 1j * rng.standard_normal(size * batch)).astype(dtype)
a = DeviceArray((size * batch,), dtype).set(data); b = DeviceArray((size * batch,), dtype)
plan = Plan(shape if len(shape) > 1 else shape[0], dtype=dtype, wait_for_finish=True)
plan.execute(a, b, batch=batch)
got = b.get()[:size].astype(numpy.complex128)
ref = numpy.fft.fftn(data[:size].reshape(shape).astype(numpy.complex128)).reshape(-1)
err = numpy.abs(got - ref).sum() / numpy.abs(ref).sum()
st = plan._context.getQueue(); best = 1e9
for _ in range(5):
    e0 = Event().record(st)
    for _ in range(10): plan.execute(a, b, batch=batch, wait_for_finish=False)
    e1 = Event().record(st); e1.synchronize(); best = min(best, e1.time_since(e0) / 10)
plan.finish()
frac = 2.0 * size * batch * dtype.itemsize / (best * 1e-3) / 8e12
print("%%-12s %%-10s x %%-3d %%-22s %%8.2f us  %%.3f of roofline  err %%.1e" %% (sys.argv[1], dtype.name, batch, str(plan.strategy(batch)[:3]), best * 1e3, frac, err))
''' % ROOT
for shape, dt in (("1048576", "complex64"), ("1024x1024", "complex64"), ("1024x1024", "complex128"), ("262144", "complex64"), ("4194304", "complex64")):
    for batch in (2, 4, 8, 16, 32):
        for env in ({"PYFFT_AMD_SMALL_FUSED": "0"}, {"PYFFT_AMD_SMALL_FUSED": "4"}, {"PYFFT_AMD_SMALL_FUSED": "2"}, {"PYFFT_AMD_SMALL_FUSED": "1"},
                    {"PYFFT_AMD_SMALL_FUSED": "0", "PYFFT_AMD_STRATEGY": "pipelined", "PYFFT_AMD_PIPE_MB": str(max(8, batch * 8 // 4))}):
            e = dict(os.environ); e.update(env)
            r = subprocess.run([sys.executable, "-c", CHILD, shape, dt, str(batch)], env=e, capture_output=True, text=True)
            print(" ".join("%s=%s" % kv for kv in env.items()).ljust(75), (r.stdout.strip() or r.stderr.strip()[-300:]), flush=True)

"""Where do the stream-pipelined chunks start to pay?  chain against pipelined (and the plan's own choice) at growing batches.
python3 tools/pipe_threshold_probe.py"""
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, numpy
sys.path.insert(0, %r)
from pyfft_amd.hip import Plan, DeviceArray, Event
from pyfft_amd import _native as N
shape = tuple(int(t) for t in sys.argv[1].split("x")); dtype = numpy.dtype(sys.argv[2]); batch = int(sys.argv[3])
size = int(numpy.prod(shape))
a = DeviceArray((size * batch,), dtype); b = DeviceArray((size * batch,), dtype)
rng = numpy.random.default_rng(5)
blk = (rng.standard_normal(size) + 1j * rng.standard_normal(size)).astype(dtype)
for i in range(batch):
    N.check(N.lib.mifft_memcpy_h2d(a.ptr + i * blk.nbytes, blk.ctypes.data, blk.nbytes, None))
plan = Plan(shape if len(shape) > 1 else shape[0], dtype=dtype, wait_for_finish=True)
plan.execute(a, b, batch=batch)
st = plan._context.getQueue(); best = 1e9
for _ in range(4):
    e0 = Event().record(st)
    for _ in range(5): plan.execute(a, b, batch=batch, wait_for_finish=False)
    e1 = Event().record(st); e1.synchronize(); best = min(best, e1.time_since(e0) / 5)
plan.finish()
frac = 2.0 * size * batch * dtype.itemsize / (best * 1e-3) / 8e12
print("%%-12s %%-10s x %%-4d %%-24s %%9.2f us  %%.3f" %% (sys.argv[1], dtype.name, batch, str(plan.strategy(batch)[:3]), best * 1e3, frac))
''' % ROOT
for shape, dt, batches in (("1024x1024", "complex64", (16, 32, 64, 128, 256)), ("1048576", "complex64", (16, 32, 48, 64)),
                           ("262144", "complex64", (64, 128, 256, 512)), ("65536", "complex64", (256, 512, 1024, 2048)),
                           ("128x128x128", "complex64", (4, 8, 16, 32, 64)), ("1024x1024", "complex128", (8, 16, 32)),
                           ("256x256x256", "complex128", (1, 2, 4, 8))):
    for batch in batches:
        for env in ({"PYFFT_AMD_STRATEGY": "chain"}, {"PYFFT_AMD_STRATEGY": "pipelined"}, {}):
            e = dict(os.environ); e.update(env)
            r = subprocess.run([sys.executable, "-c", CHILD, shape, dt, str(batch)], env=e, capture_output=True, text=True)
            print((env.get("PYFFT_AMD_STRATEGY", "auto")).ljust(10), (r.stdout.strip() or r.stderr.strip()[-300:]), flush=True)

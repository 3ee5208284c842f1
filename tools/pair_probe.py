"""Pass-pair kernels (csrc/fft_pair.hpp) on 256^3 complex128: parity against numpy (out of place, in place, inverse) and the time
per execute next to the one-pass-per-axis chain.  python3 tools/pair_probe.py [batch]   (MIFFT_PAIR=1 off, 2 alternative split)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
from pyfft_amd.hip import Plan, DeviceArray, Event
from pyfft_amd import _native as N

shape = (256, 256, 256)
size = 1 << 24
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
plan = Plan(shape, dtype=numpy.complex128, wait_for_finish=True)
print("passes", plan.pass_list(), "strategy", plan.strategy(B), flush=True)
rng = numpy.random.default_rng(1)
data = (rng.standard_normal((2,) + shape) + 1j * rng.standard_normal((2,) + shape)).astype(numpy.complex128)
ref = numpy.fft.fftn(data, axes=(1, 2, 3))
a = DeviceArray((2 * size,), numpy.complex128).set(data.reshape(-1))
b = DeviceArray((2 * size,), numpy.complex128)
plan.execute(a, b, batch=2)
got = b.get().reshape(ref.shape)
print("out of place: L1-rel %.3e  max-rel %.3e  input untouched %s" % (
    numpy.abs(got - ref).sum() / numpy.abs(ref).sum(), numpy.abs(got - ref).max() / numpy.abs(ref).max(),
    numpy.array_equal(a.get().reshape(data.shape), data)), flush=True)
plan.execute(a, batch=2)
got2 = a.get().reshape(ref.shape)
print("in place:     L1-rel %.3e  identical to out of place %s" % (numpy.abs(got2 - ref).sum() / numpy.abs(ref).sum(), numpy.array_equal(got, got2)), flush=True)
plan.execute(a, batch=2, inverse=True)
back = a.get().reshape(data.shape)
print("inverse:      L1-rel %.3e" % (numpy.abs(back - data).sum() / numpy.abs(data).sum()), flush=True)
del a, b
x = DeviceArray((B * size,), numpy.complex128)
y = DeviceArray((B * size,), numpy.complex128)
N.check(N.lib.mifft_memset(x.ptr, 0, x.nbytes, None), "memset")
for name, args in (("out of place", (x, y)), ("in place", (x,))):
    plan.execute(*args, batch=B)
    st = plan._context.getQueue()
    e0 = Event().record(st)
    for _ in range(5):
        plan.execute(*args, batch=B, wait_for_finish=False)
    e1 = Event().record(st)
    e1.synchronize()
    plan.finish()
    ms = e1.time_since(e0) / 5
    print("%-12s batch %d: %.3f ms  %.1f %% of 8 TB/s  (%s)" % (name, B, ms, 2.0 * size * 16 * B / (ms * 1e-3) / 8e12 * 100, plan.strategy(B)[0]), flush=True)

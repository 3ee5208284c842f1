try:
    import torch
except Exception:
    pass
import sys, os, gc
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy
from pyfft_amd import hip
from pyfft_amd import _native as N
from helpers import _tiled_noise
shape, batch, dtype = (128, 512, 512), 2, numpy.complex64
size = int(numpy.prod(shape))
data = _tiled_noise(size * batch, dtype, 501)
s = hip.Stream()
plan = hip.Plan(shape, dtype=dtype, stream=s)
a = hip.to_gpu(data); b = hip.DeviceArray((size * batch,), dtype)
print(plan.strategy(batch))
plan.execute(a, b, batch=batch); s.synchronize()
want = b.get().view(numpy.uint32)
def check(tag):
    N.check(N.lib.mifft_memset(b.ptr, 0, b.nbytes, s.handle))
    g.launch(); s.synchronize()
    got = b.get().view(numpy.uint32)
    bad = numpy.nonzero(got != want)[0]
    print(tag, "mismatching words:", bad.size, (bad[:4], bad[-4:]) if bad.size else "", flush=True)
with hip.Graph(s) as g:
    plan.execute(a, b, batch=batch)
check("after capture")
for i in range(5):
    N.check(N.lib.mifft_memset(b.ptr, 0, b.nbytes, s.handle))
    g.launch()
    if i in (1, 2):
        g.launch()
    s.synchronize()
    assert numpy.array_equal(b.get().view(numpy.uint32), want), ("replay", i)
    if i in (0, 3):
        N.check(N.lib.mifft_memset(b.ptr, 0, b.nbytes, s.handle))
        plan.execute(a, b, batch=batch)
        if i == 3:
            plan.execute(a, b, batch=batch)
        s.synchronize()
        assert numpy.array_equal(b.get().view(numpy.uint32), want), ("eager after replay", i)
plan.finish()
check("after the loop")
os.environ["PYFFT_AMD_NO_PLANE_FUSED"] = "1"
junk = hip.DeviceArray((1 << 20,), numpy.uint8)
check("after a device allocation")
junk.set(numpy.zeros(1 << 20, numpy.uint8))
check("after an h2d copy")
other = hip.Plan(shape, dtype=dtype, stream=s)
check("after creating the other plan")
print("other", other.strategy(batch))
check("after other.strategy")
other.execute(a, b, batch=batch)
s.synchronize()
del os.environ["PYFFT_AMD_NO_PLANE_FUSED"]
check("after the other plan's execute")
sys.exit(0)
with hip.Graph(s) as g2:
    plan.execute(a, b, batch=batch - 1)
check("after capturing batch-1")
N.check(N.lib.mifft_memset(b.ptr, 0, b.nbytes, s.handle))
g2.launch(); s.synchronize()
check("after launching the batch-1 graph")
try:
    with hip.Graph(s):
        plan.execute(a, b, batch=batch, wait_for_finish=True)
except RuntimeError as e:
    print("expected:", str(e)[:80])
check("after the refused capture")
plan.execute(a, b, batch=1)
check("after eager batch 1")
plan.close()
check("after close")
del plan; gc.collect()
check("after del")

"""Does building ANOTHER plan invalidate a graph recorded from a plan?  (Seen under the HIP runtime torch bundles, with the plane-fused
route; run with and without `import torch` first.)"""
import os, sys
if os.environ.get("WITH_TORCH"):
    import torch  # noqa: F401
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy
from pyfft_amd import hip
from pyfft_amd import _native as N
from helpers import _tiled_noise

for shape, batch in (((1 << 20,), 64), ((128, 128, 128), 32), ((1 << 16,), 96), ((1024, 1024), 64), ((128, 512, 512), 2), ((256, 4096), 40)):
    size = int(numpy.prod(shape))
    data = _tiled_noise(size * batch, numpy.complex64, 501)
    s = hip.Stream()
    plan = hip.Plan(shape if len(shape) > 1 else shape[0], dtype=numpy.complex64, stream=s)
    a = hip.to_gpu(data); b = hip.DeviceArray((size * batch,), numpy.complex64)
    plan.execute(a, b, batch=batch); s.synchronize()
    want = b.get().view(numpy.uint32)
    with hip.Graph(s) as g:
        plan.execute(a, b, batch=batch)
    def check():
        N.check(N.lib.mifft_memset(b.ptr, 0, b.nbytes, s.handle))
        g.launch(); s.synchronize()
        return int(numpy.count_nonzero(b.get().view(numpy.uint32) != want))
    r = [check()]
    p2 = hip.Plan(64, dtype=numpy.complex64, stream=s)              # a small unrelated plan
    r.append(check())
    p3 = hip.Plan(shape if len(shape) > 1 else shape[0], dtype=numpy.complex64, stream=s)     # the same shape again
    r.append(check())
    p4 = hip.Plan(shape if len(shape) > 1 else shape[0], dtype=numpy.complex64)               # ... without a stream
    r.append(check())
    print(shape, batch, plan.strategy(batch)[0], "mismatching words after [capture, small plan, same-shape plan, plan without stream]:", r, flush=True)
    del a, b, plan, g, p2, p3, p4

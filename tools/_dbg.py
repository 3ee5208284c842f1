import os, sys, numpy
sys.path.insert(0, "."); sys.path.insert(0, "oracle"); sys.path.insert(0, "tests")
import torch
from helpers import HipContext
import pyfft_oracle as oracle
ctx = HipContext()
n, batch = 1 << 20, 31
data = oracle.get_test_data((n,), numpy.complex128, batch, 4243)
ref = None
def rep(tag, got):
    bad = numpy.nonzero(got != ref)[0]
    print(tag, "mismatches", bad.size)
    if bad.size:
        print("  transforms", numpy.unique(bad // n)[:40], "\n  rows", numpy.unique((bad % n) // 1024).size, numpy.unique((bad % n) // 1024)[:70], "\n  cols", numpy.unique(bad % 1024)[:40])
        k = bad[0]; print("  first", k, got[k], ref[k])
for strat in ("chain", "fused"):
    os.environ["PYFFT_AMD_STRATEGY"] = strat
    plan = ctx.getPlan((n,), dtype=numpy.complex128)
    a = ctx.toGpu(data); b = ctx.allocate(data.shape, data.dtype)
    plan.execute(a, b, batch=batch)
    o = b.get()
    if ref is None: ref = o
    rep(strat + " outp", o)
    a = ctx.toGpu(data)
    plan.execute(a, batch=batch)
    rep(strat + " inpl", a.get())

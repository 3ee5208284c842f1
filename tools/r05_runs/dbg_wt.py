"""debug: nd2t split kernel with write-through stores (round 5)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy
from pyfft_amd import hip
rng = numpy.random.default_rng(1)
for shape, dt in (((16, 16), numpy.float64), ((64, 64), numpy.float32)):
    batch = 64
    re = rng.standard_normal((batch,) + shape).astype(dt); im = rng.standard_normal((batch,) + shape).astype(dt)
    plan = hip.Plan(shape, dtype=dt, wait_for_finish=True)
    a_re, a_im = hip.to_gpu(re.reshape(-1)), hip.to_gpu(im.reshape(-1))
    b_re, b_im = hip.DeviceArray(a_re.shape, dt), hip.DeviceArray(a_im.shape, dt)
    plan.execute(a_re, a_im, b_re, b_im, batch=batch)
    got = (b_re.get() + 1j * b_im.get()).reshape((batch,) + shape)
    ref = numpy.fft.fftn(re + 1j * im, axes=(1, 2))
    d = numpy.abs(got - ref)
    print(shape, numpy.dtype(dt).name, "rel err", d.sum() / numpy.abs(ref).sum(), "zeros in re", int((b_re.get() == 0).sum()), "of", b_re.get().size)
    bad = numpy.argwhere(d > 1e-3 * numpy.abs(ref).max())
    print("  bad points", len(bad), "first", bad[:6].tolist())
    print("  re ok / im ok:", float(numpy.abs(got.real - ref.real).max()), float(numpy.abs(got.imag - ref.imag).max()))

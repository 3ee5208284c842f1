"""Seeded randomized sweep through the public API on the GPU: random power-of-two shapes (1-3 axes), ragged
batches (tile tails), four dtypes, forward/inverse, in/out of place, normalize and scale -- every case against
numpy.fft on the complex128-upcast input with the reference's thresholds (test/test_errors.py:20-23)."""
import os

import numpy
import pytest

import pyfft_oracle as oracle

pytestmark = pytest.mark.gpu


def _cases(seed, count, max_points):
    rng = numpy.random.default_rng(seed)
    out = []
    while len(out) < count:
        ndim = int(rng.integers(1, 4))
        logs = [int(rng.integers(1, 12)) for _ in range(ndim)]
        if sum(logs) > max_points:
            continue
        shape = tuple(1 << l for l in logs)
        size = int(numpy.prod(shape))
        batch = int(rng.integers(1, max(2, min(70, (1 << max(18, max_points + 1)) // size))))
        dtype = [numpy.complex64, numpy.float32, numpy.complex128, numpy.float64][int(rng.integers(0, 4))]
        out.append((shape, batch, dtype, bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2)),
                    float([1.0, 0.5, 3.0][int(rng.integers(0, 3))])))
    return out


# PYFFT_AMD_SWEEP=<count>[:<seed>[:<max log2 points>]] widens the sweep for a one-off soak run (default: 60 cases)
_SWEEP = os.environ.get("PYFFT_AMD_SWEEP", "60:20261002").split(":")
_COUNT, _SEED = int(_SWEEP[0]), int(_SWEEP[1]) if len(_SWEEP) > 1 else 20261002
_MAXLOG = int(_SWEEP[2]) if len(_SWEEP) > 2 else 17


@pytest.mark.parametrize("case", _cases(_SEED, _COUNT, _MAXLOG), ids=lambda c: "%s-b%d-%s-%s%s" % (
    "x".join(map(str, c[0])), c[1], numpy.dtype(c[2]).name, "inv" if c[3] else "fwd", "-ip" if c[4] else ""))
def test_random_case(ctx, case):
    shape, batch, dtype, inverse, inplace, normalize, scale = case
    dtype = numpy.dtype(dtype)
    split = dtype.kind == "f"
    double = dtype in (numpy.dtype(numpy.float64), numpy.dtype(numpy.complex128))
    eps, mx = (1e-11, 1e-10) if double else (1.1e-6, 1e-5)
    seed = (sum((i + 1) * 7919 * n for i, n in enumerate(shape)) + 31 * batch + dtype.itemsize) % (1 << 31)
    if split:
        re, im = oracle.get_test_data(shape, dtype, batch, seed)
        data = re.astype(numpy.complex128) + 1j * im
    else:
        data = oracle.get_test_data(shape, dtype, batch, seed)
    size = int(numpy.prod(shape))
    ref = oracle.numpy_fft(numpy.fft.ifftn if inverse else numpy.fft.fftn, data, batch)
    if inverse:
        ref = ref * (size if not normalize else 1.0) / scale      # kernel.py:33-35
    else:
        ref = ref * scale                                         # kernel.py:31-32
    plan = ctx.getPlan(shape, dtype=dtype, normalize=normalize, scale=scale)
    if split:
        a, b = ctx.toGpu(re), ctx.toGpu(im)
        if inplace:
            plan.execute(a, b, batch=batch, inverse=inverse)
            got = a.get().astype(numpy.complex128) + 1j * b.get()
        else:
            c, d = ctx.allocate(re.shape, dtype), ctx.allocate(im.shape, dtype)
            plan.execute(a, b, c, d, batch=batch, inverse=inverse)
            got = c.get().astype(numpy.complex128) + 1j * d.get()
            assert numpy.array_equal(a.get(), re) and numpy.array_equal(b.get(), im)
    else:
        a = ctx.toGpu(data)
        if inplace:
            plan.execute(a, batch=batch, inverse=inverse)
            got = a.get().astype(numpy.complex128)
        else:
            c = ctx.allocate(data.shape, dtype)
            plan.execute(a, c, batch=batch, inverse=inverse)
            got = c.get().astype(numpy.complex128)
            assert numpy.array_equal(a.get(), data)
    assert oracle.difference(ref, got.reshape(ref.shape), batch) < eps
    assert numpy.abs(ref - got.reshape(ref.shape)).max() <= mx * numpy.abs(ref).max()

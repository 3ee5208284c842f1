"""Parity at BASELINE.json's full sizes through size-independent properties (the oracle cannot run these sizes
in seconds): sampled batch items against numpy.fft on the complex128-upcast input, forward->inverse round trip,
linearity, and in-place == out-of-place, for configs 2, 3, 4 (both layouts) and the per-GPU chunk of config 5.
The input is one host block of <= 32 transforms tiled across the batch on the device."""
import numpy
import pytest

pytestmark = pytest.mark.gpu

EPS = {"f": (1.1e-6, 1e-5), "d": (1e-11, 1e-10)}   # (L1-relative difference, max-norm) -- test_errors.py:20-23 + north star


def _tile(N, dst, nbytes, host):
    hb = host.view(numpy.uint8).reshape(-1)
    n0 = min(nbytes, hb.nbytes)
    N.check(N.lib.mifft_memcpy_h2d(dst, hb.ctypes.data, n0, None))
    done = n0
    while done < nbytes:
        n = min(done, nbytes - done)
        N.check(N.lib.mifft_memcpy_d2d(dst + done, dst, n, None))
        done += n
    N.check(N.lib.mifft_device_sync())


def _fetch(N, ptr, item, size, dtype):
    out = numpy.empty(size, dtype)
    N.check(N.lib.mifft_memcpy_d2h(out.ctypes.data, ptr + item * size * out.itemsize, out.nbytes, None))
    return out


def _full_size(ctx, shape, dtype, batch, seed):
    from pyfft_amd import _native as N
    hip = ctx.hip
    dtype = numpy.dtype(dtype)
    split = dtype.kind == "f"
    prec = "f" if dtype in (numpy.complex64, numpy.float32) else "d"
    cdt = numpy.complex64 if prec == "f" else numpy.complex128
    fdt = numpy.float32 if prec == "f" else numpy.float64
    size = int(numpy.prod(shape))
    blk = max(1, min(batch, 32, (256 << 20) // (size * numpy.dtype(cdt).itemsize)))
    rng = numpy.random.default_rng(seed)
    h_re = rng.standard_normal((blk, size)).astype(fdt)
    h_im = rng.standard_normal((blk, size)).astype(fdt)
    host_c = (h_re + 1j * h_im).astype(cdt)
    nel = size * batch
    planes = 2 if split else 1
    a = [hip.DeviceArray((nel,), dtype) for _ in range(planes)]
    b = [hip.DeviceArray((nel,), dtype) for _ in range(planes)]
    if split:
        _tile(N, a[0].ptr, a[0].nbytes, h_re)
        _tile(N, a[1].ptr, a[1].nbytes, h_im)
    else:
        _tile(N, a[0].ptr, a[0].nbytes, host_c)
    plan = hip.Plan(shape if len(shape) > 1 else shape[0], dtype=dtype)

    def get(bufs, item):
        if split:
            return _fetch(N, bufs[0].ptr, item, size, dtype).astype(numpy.complex128) + \
                1j * _fetch(N, bufs[1].ptr, item, size, dtype)
        return _fetch(N, bufs[0].ptr, item, size, dtype).astype(numpy.complex128)

    def execute(src, dst, **kw):
        if split:
            if dst is None:
                plan.execute(src[0], src[1], batch=batch, **kw)
            else:
                plan.execute(src[0], src[1], dst[0], dst[1], batch=batch, **kw)
        else:
            if dst is None:
                plan.execute(src[0], batch=batch, **kw)
            else:
                plan.execute(src[0], dst[0], batch=batch, **kw)

    eps, mx = EPS[prec]
    samples = sorted(set([0, 1 % batch, blk - 1, blk % batch, batch // 2 + 1, batch - 1]))
    # forward, out of place: sampled items vs numpy; input untouched
    execute(a, b)
    for s in samples:
        ref = numpy.fft.fftn(host_c[s % blk].astype(numpy.complex128).reshape(shape)).reshape(-1)
        got = get(b, s)
        assert numpy.abs(ref - got).sum() / numpy.abs(ref).sum() < eps, (shape, s)
        assert numpy.abs(ref - got).max() <= mx * numpy.abs(ref).max(), (shape, s)
        assert numpy.array_equal(get(a, s), host_c[s % blk].astype(numpy.complex128)), "input modified"
    # periodic input -> periodic output (every copy of a host item must transform identically): checks that no
    # batch item is skipped or mixed up anywhere in the batch
    if batch > blk:
        first = get(b, 0)
        for s in (blk, (batch // blk - 1) * blk):
            assert numpy.array_equal(get(b, s), first), "batch items with identical input differ"
    # inverse in place on the result: round trip
    execute(b, None, inverse=True)
    for s in samples:
        want = host_c[s % blk].astype(numpy.complex128)
        got = get(b, s)
        assert numpy.abs(want - got).sum() / numpy.abs(want).sum() < eps, (shape, s)
    # in place forward == out of place forward (b now holds the input again up to rounding: use a instead)
    execute(a, b)
    keep = [get(b, s) for s in samples]
    execute(a, None)
    for s, k in zip(samples, keep):
        assert numpy.array_equal(get(a, s), k), "in-place result differs from out-of-place"


def test_config2_full(ctx):
    """1-D c2c fp32 N = 2^20, batch 4096 (32 GiB in + 32 GiB out + scratch)."""
    _full_size(ctx, (1 << 20,), numpy.complex64, 4096, 1002)


def test_config3_full(ctx):
    """2-D c2c fp32 1024 x 1024, batch 512."""
    _full_size(ctx, (1024, 1024), numpy.complex64, 512, 1003)


@pytest.mark.parametrize("dtype", [numpy.complex128, numpy.float64], ids=["interleaved", "split"])
def test_config4_full(ctx, dtype):
    """3-D c2c fp64 256^3, batch 64, interleaved and split-complex."""
    _full_size(ctx, (256, 256, 256), dtype, 64, 1004)


def test_config5_per_gpu_chunk(ctx):
    """1-D c2c fp32 N = 2^22: a resident chunk (256 transforms = 8 GiB) of the 8192 per GPU of config 5."""
    _full_size(ctx, (1 << 22,), numpy.complex64, 256, 1005)


def test_linearity_large(ctx):
    """F(a*x + b*y) == a*F(x) + b*F(y) at N = 2^20 (fp32 tolerance on the L1 metric)."""
    hip = ctx.hip
    n, batch = 1 << 20, 4
    rng = numpy.random.default_rng(9)
    x = (rng.standard_normal(n * batch) + 1j * rng.standard_normal(n * batch)).astype(numpy.complex64)
    y = (rng.standard_normal(n * batch) + 1j * rng.standard_normal(n * batch)).astype(numpy.complex64)
    z = (numpy.complex64(2.5) * x - numpy.complex64(1j) * y).astype(numpy.complex64)
    plan = hip.Plan(n, dtype=numpy.complex64)
    res = []
    for v in (x, y, z):
        g = hip.to_gpu(v)
        plan.execute(g, batch=batch)
        res.append(g.get().astype(numpy.complex128))
    lin = 2.5 * res[0] - 1j * res[1]
    assert numpy.abs(lin - res[2]).sum() / numpy.abs(lin).sum() < 2e-6

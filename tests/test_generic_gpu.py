"""GPU tests of the opt-in extensions the reference lists as TODO (TODO.txt:6-8; pyfft_amd/generic.py): sizes that are not powers of two
(mixed-radix rows, long smooth lengths, Bluestein in one launch, smooth N-D shapes) and tiled batches (tiles of a parent array in one launch,
interleaved and split-complex).  numpy.fft on the complex128-upcast input with the reference's thresholds (test/test_errors.py:20-23)."""
import ctypes
import json
import os
import subprocess
import sys

import numpy
import pytest

import pyfft_oracle as oracle
from helpers import EPS_F, MAX_F, getDimensions, _execute, _execute_split, _noise, _test_data, _tiled_noise
from test_errors_gpu import run_protocol

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- f4: the reference's TODO items (TODO.txt:6-8), opt-in extensions of Plan() (pyfft_amd/generic.py) ---------------------
def _numpy_tiles(x, batch, shape, parent):
    """numpy.fft.fftn of every non-overlapping tile of `shape` in `batch` parent arrays"""
    import itertools
    pshape = parent if parent else shape
    xb = x.reshape((batch,) + tuple(pshape)).astype(numpy.complex128)
    ref = numpy.empty_like(xb)
    counts = [p // t for p, t in zip(pshape, shape)]
    for bi in range(batch):
        for idx in itertools.product(*[range(c) for c in counts]):
            sl = tuple(slice(i * t, (i + 1) * t) for i, t in zip(idx, shape))
            ref[(bi,) + sl] = numpy.fft.fftn(xb[(bi,) + sl])
    return ref.reshape(x.shape)


def _run_generic(ctx, shape, dtype, batch, parent=None, seed=0):
    """Forward out of place (input untouched) against numpy with the reference's thresholds (test/test_errors.py:20-23) and the
    north star's max-norm bound, then the normalised inverse in place back to the input."""
    dt = numpy.dtype(dtype)
    double = dt in (numpy.dtype(numpy.complex128), numpy.dtype(numpy.float64))
    cd = numpy.complex128 if double else numpy.complex64
    split = dt.kind == "f"
    eps, mx = (1e-11, 1e-10) if double else (1.1e-6, 1e-5)
    pshape = parent if parent else shape
    full = (batch * pshape[0],) + tuple(pshape[1:])
    rng = numpy.random.default_rng(900 + seed)
    x = (rng.standard_normal(full) + 1j * rng.standard_normal(full)).astype(cd)
    plan = ctx.getPlan(shape, dtype=dtype, any_size=True, parent_shape=parent)
    ref = _numpy_tiles(x, batch, shape, parent)
    if split:
        a = [ctx.toGpu(numpy.ascontiguousarray(x.real)), ctx.toGpu(numpy.ascontiguousarray(x.imag))]
        b = [ctx.allocate(full, dt), ctx.allocate(full, dt)]
        plan.execute(a[0], a[1], b[0], b[1], batch=batch)
        got = b[0].get() + 1j * b[1].get()
        assert numpy.array_equal(a[0].get() + 1j * a[1].get(), x)
        plan.execute(b[0], b[1], inverse=True, batch=batch)
        back = b[0].get() + 1j * b[1].get()
    else:
        a, b = ctx.toGpu(x), ctx.allocate(full, cd)
        plan.execute(a, b, batch=batch)
        got = b.get()
        assert numpy.array_equal(a.get(), x)
        plan.execute(b, inverse=True, batch=batch)
        back = b.get()
    assert numpy.abs(got - ref).sum() / numpy.abs(ref).sum() < eps
    assert numpy.abs(got - ref).max() <= mx * numpy.abs(ref).max()
    assert numpy.abs(back - x).sum() / numpy.abs(x).sum() < 2 * eps


@pytest.mark.parametrize("dtype", [numpy.complex64, numpy.complex128, numpy.float32, numpy.float64], ids=["c64", "c128", "f32", "f64"])
@pytest.mark.parametrize("shape", [(3,), (5,), (7,), (12,), (100,), (1000,), (1023,), (4097,), (30000,), (12, 20), (100, 64),
                                   (64, 100), (6, 10, 14), (16, 9, 32), (3, 1, 5)], ids=str)
def test_any_size_plans(ctx, shape, dtype):
    """Sizes that are not powers of two (TODO.txt:8), 1-D / 2-D / 3-D, mixed with power-of-two axes: Bluestein per axis."""
    _run_generic(ctx, shape, dtype, 3, seed=sum(shape))


@pytest.mark.parametrize("shape,parent,dtype", [((16, 16), (64, 64), numpy.complex64), ((4, 16, 32), (8, 32, 64), numpy.complex128),
                                                ((256,), (1024,), numpy.float32), ((32, 8), (32, 64), numpy.float64),
                                                ((12, 10), (24, 30), numpy.complex64), ((128, 128), (256, 512), numpy.complex64)], ids=str)
def test_tiled_batches(ctx, shape, parent, dtype):
    """2D/3D tiled batches (TODO.txt:6-7): every non-overlapping tile of a bigger array in one execute()."""
    _run_generic(ctx, shape, dtype, 2, parent=parent, seed=sum(parent))


def test_any_size_is_opt_in_and_keeps_the_reference_errors(ctx):
    with pytest.raises(ValueError):
        ctx.getPlan((17,), dtype=numpy.complex64)                       # the reference's behaviour (plan.py:23-24)
    with pytest.raises(ValueError):
        ctx.getPlan((16, 16), dtype=numpy.complex64, parent_shape=(40, 64))   # not a multiple of the tile
    with pytest.raises(ValueError):
        ctx.getPlan((16, 16), dtype=numpy.complex64, parent_shape=(64,))      # rank mismatch
    with pytest.raises(ValueError):
        ctx.getPlan((17,), dtype=numpy.int32, any_size=True)
    # normalize / scale semantics carry over (test/test_functionality.py:53-100)
    data = numpy.ones(15, dtype=numpy.complex64)
    for normalize in (True, False):
        plan = ctx.getPlan((15,), dtype=numpy.complex64, any_size=True, normalize=normalize, scale=10.0)
        g = ctx.toGpu(data)
        plan.execute(g)
        assert numpy.abs(g.get() - numpy.fft.fft(data) * 10.0).max() < 1e-4
        plan.execute(g, inverse=True)
        assert numpy.abs(g.get() - data * (1 if normalize else 15)).max() < 1e-4
    # asynchronous form returns the stream
    s = ctx.hip.Stream()
    plan = ctx.getPlan((15,), dtype=numpy.complex64, any_size=True, stream=s)
    assert plan.execute(ctx.toGpu(data)) is s
    s.synchronize()


# ---- tiled batches in one launch (csrc/fft_nd2t.hpp; the reference's TODO.txt:6-7) -------------------------------------
TILED = [((8, 8), (24, 40)), ((16, 16), (48, 80)), ((32, 32), (96, 64)), ((64, 64), (192, 128)), ((128, 128), (256, 384)),
         ((16, 32), (32, 96)), ((32, 64), (96, 64)), ((64, 128), (128, 384)), ((8, 8, 8), (16, 24, 8)), ((16, 16, 16), (32, 16, 48)),
         ((8, 16, 16), (8, 48, 32)), ((8, 32, 32), (24, 32, 64))]


@pytest.mark.parametrize("dtype", [numpy.complex64, numpy.complex128], ids=["c64", "c128"])
@pytest.mark.parametrize("shape,parent", TILED + [((32, 32, 32), (64, 32, 96))], ids=str)
def test_tiled_batch_single_launch(ctx, shape, parent, dtype, monkeypatch):
    """Every tile shape of the tiled N-D kernel: the tiles of 3 parent arrays transformed where they lie (one launch, no work
    array), tile counts that leave the last work-group ragged -- against numpy tile by tile (reference thresholds), against the
    gather / dense plan / scatter form of the same plan, out of place with the input untouched, in place, and the inverse."""
    from test_generic_gpu import _numpy_tiles
    if shape == (32, 32, 32) and numpy.dtype(dtype) == numpy.complex128:
        shape, parent = (16, 32, 32), (32, 32, 96)            # fp64: the largest cube-like tile is (z, y, x) = (16, 32, 32)
    batch = 3
    cd = numpy.dtype(dtype)
    eps, mx = (1e-11, 1e-10) if cd == numpy.complex128 else (1.1e-6, 1e-5)
    full = (batch * parent[0],) + tuple(parent[1:])
    rng = numpy.random.default_rng(77 + sum(parent))
    x = (rng.standard_normal(full) + 1j * rng.standard_normal(full)).astype(cd)
    ref = _numpy_tiles(x, batch, shape, parent)
    plan = ctx.getPlan(shape, dtype=dtype, parent_shape=parent)
    assert plan._tiled and plan._work is None
    a, b = ctx.toGpu(x), ctx.allocate(full, cd)
    plan.execute(a, b, batch=batch)
    got = b.get()
    assert numpy.array_equal(a.get(), x), "out-of-place execute modified its input"
    assert numpy.abs(got - ref).sum() / numpy.abs(ref).sum() < eps
    assert numpy.abs(got - ref).max() <= mx * numpy.abs(ref).max()
    c = ctx.toGpu(x)
    plan.execute(c, batch=batch)                              # in place
    assert numpy.array_equal(c.get(), got)
    plan.execute(c, inverse=True, batch=batch)
    assert numpy.abs(c.get() - x).sum() / numpy.abs(x).sum() < 2 * eps
    assert plan._work is None                                 # never needed a work array
    # the three-round-trip form of the same plan (round 4: a one-launch plan no longer builds the inner N-D plan it does not run)
    monkeypatch.setenv("PYFFT_AMD_NO_TILED", "1")
    plan3 = ctx.getPlan(shape, dtype=dtype, parent_shape=parent)
    assert not plan3._tiled and len(plan3._inner_plans()) == 1 and plan._inner_plans() == []
    d = ctx.allocate(full, cd)
    plan3.execute(a, d, batch=batch)
    assert numpy.abs(d.get() - got).sum() / numpy.abs(got).sum() < eps


# ---- mixed-radix rows for smooth lengths (csrc/fft_mixed.hip; the reference's TODO.txt:8) -------------------------------
@pytest.mark.parametrize("dtype", [numpy.complex64, numpy.complex128], ids=["c64", "c128"])
@pytest.mark.parametrize("n", [2, 3, 5, 6, 7, 9, 10, 12, 15, 21, 25, 27, 35, 49, 60, 64, 100, 105, 125, 210, 243, 343, 360, 625, 1000,
                               1029, 1200, 2000, 2048, 2187, 2401, 3125, 3600, 4000, 4096])
def test_mixed_radix_rows(ctx, n, dtype):
    """Rows of smooth length through the C ABI: forward out of place with padded rows, inverse in place, against numpy with the
    reference's thresholds; lengths the kernel does not take are refused."""
    import ctypes
    from pyfft_amd import _native as N
    cd = numpy.dtype(dtype)
    prec = N.F64 if cd == numpy.complex128 else N.F32
    if N.lib.mifft_mixed_supported(prec, n) != 0:
        assert prec == N.F64 and n > 2048
        return
    eps, mx = (1e-11, 1e-10) if prec == N.F64 else (1.1e-6, 1e-5)
    rows, pad = 37, 4
    rng = numpy.random.default_rng(n)
    x = (rng.standard_normal((rows, n + pad)) + 1j * rng.standard_normal((rows, n + pad))).astype(cd)
    k = numpy.arange(n, dtype=numpy.float64)
    tw = ctx.toGpu(numpy.exp(-2j * numpy.pi * k / n).astype(cd))
    a = ctx.toGpu(x)
    b = ctx.allocate((rows, n), cd)
    N.check(N.lib.mifft_launch_mixed_rows(prec, n, rows, n + pad, n, a.ptr, b.ptr, tw.ptr, 0, 2.0, None), "mixed")
    N.check(N.lib.mifft_device_sync(), "sync")
    ref = 2.0 * numpy.fft.fft(x[:, :n].astype(numpy.complex128), axis=1)
    got = b.get()
    assert numpy.abs(got - ref).sum() / numpy.abs(ref).sum() < eps
    assert numpy.abs(got - ref).max() <= mx * numpy.abs(ref).max()
    assert numpy.array_equal(a.get(), x)
    N.check(N.lib.mifft_launch_mixed_rows(prec, n, rows, n, n, b.ptr, b.ptr, tw.ptr, 1, 0.5 / n, None), "mixed")
    N.check(N.lib.mifft_device_sync(), "sync")
    back = b.get()
    assert numpy.abs(back - x[:, :n]).sum() / numpy.abs(x[:, :n]).sum() < 2 * eps


def test_mixed_radix_is_what_smooth_any_size_plans_run(ctx):
    """Plan(shape, any_size=True): smooth axes take the mixed-radix rows, other lengths Bluestein, powers of two the batched plans."""
    plan = ctx.getPlan((1000, 17, 64), dtype=numpy.complex64, any_size=True)
    kinds = [("mixed" if ax.mixed_tw is not None else "pow2" if ax.pow2 else "bluestein") for ax in plan._axes]   # x, y, z
    assert kinds == ["pow2", "bluestein", "mixed"]
    assert ctx.hip.N.lib.mifft_mixed_supported(0, 1023) != 0 and ctx.hip.N.lib.mifft_mixed_supported(0, 8192) != 0
    assert plan._direct_nd is None and not plan._direct_mixed          # (an axis needs Bluestein: gathers stay)
    # every axis smooth: one launch per axis on the user's buffers, no work array -- against numpy, out of place and in place
    for shape in ((100, 64), (6, 10, 14), (49, 125)):
        p2 = ctx.getPlan(shape, dtype=numpy.complex128, any_size=True)
        assert p2._direct_nd is not None
        x = oracle.get_test_data(shape, numpy.complex128, 3, 21)
        a, b = ctx.toGpu(x), ctx.allocate(x.shape, x.dtype)
        p2.execute(a, b, batch=3)
        ref = oracle.numpy_fft(numpy.fft.fftn, x, 3)
        assert oracle.difference(ref, b.get(), 3) < 1e-11 and numpy.array_equal(a.get(), x)
        p2.execute(a, batch=3)
        assert numpy.array_equal(a.get(), b.get())
        p2.execute(a, batch=3, inverse=True)
        assert oracle.difference(x, a.get(), 3) < 1e-11
        assert p2._work is None
    assert ctx.getPlan((1000,), dtype=numpy.complex64, any_size=True)._direct_mixed


@pytest.mark.parametrize("dtype", [numpy.complex64, numpy.complex128], ids=["c64", "c128"])
@pytest.mark.parametrize("n", [4800, 5000, 6000, 10000, 30000, 50000, 40960, 196608, 2100 * 3])
def test_long_smooth_lengths_two_launches(ctx, n, dtype):
    """Smooth lengths beyond one tile of the mixed-radix kernel: n = n1 * n2 in two launches (lines of n1 stored as rows and
    twiddled, lines of n2), no Bluestein -- against numpy with the reference's thresholds, out of place (input untouched), in place
    (bit-identical to out of place), the normalised inverse, a ragged batch; the scale."""
    cd = numpy.dtype(dtype)
    double = cd == numpy.complex128
    eps, mx = (1e-11, 1e-10) if double else (1.1e-6, 1e-5)
    if n <= (2048 if double else 4096):
        pytest.skip("one tile")
    plan = ctx.getPlan((n,), dtype=dtype, any_size=True, scale=3.0)
    assert plan._direct_long is not None and plan._direct_long[0] * plan._direct_long[1] == n
    batch = 7
    rng = numpy.random.default_rng(n)
    x = (rng.standard_normal((batch, n)) + 1j * rng.standard_normal((batch, n))).astype(cd)
    ref = 3.0 * numpy.fft.fft(x.astype(numpy.complex128), axis=1)
    a, b = ctx.toGpu(x), ctx.allocate(x.shape, cd)
    plan.execute(a, b, batch=batch)
    got = b.get()
    assert numpy.abs(got - ref).sum() / numpy.abs(ref).sum() < eps
    assert numpy.abs(got - ref).max() <= mx * numpy.abs(ref).max()
    assert numpy.array_equal(a.get(), x) and plan._work is None
    plan.execute(a, batch=batch)                                   # in place: through the scratch array
    assert numpy.array_equal(a.get(), got) and plan._work is not None
    plan.execute(a, batch=batch, inverse=True)
    assert numpy.abs(a.get() - x).sum() / numpy.abs(x).sum() < 2 * eps
    c = ctx.toGpu(x[:3])
    plan.execute(c, batch=3)                                       # another batch: scratch re-sized on demand
    assert numpy.array_equal(c.get(), got[:3])


@pytest.mark.parametrize("dtype", [numpy.complex64, numpy.complex128], ids=["c64", "c128"])
@pytest.mark.parametrize("n", [2, 3, 11, 17, 97, 127, 513, 1009, 1023, 1100, 2039, 2048])
def test_bluestein_in_one_launch(ctx, n, dtype):
    """Rows of ANY length whose padded length fits a tile: Bluestein's algorithm in one launch (both m-point transforms in LDS).
    The C entry point with padded row strides against numpy (reference thresholds), input untouched, in place, inverse; then the
    any_size plan, which must take this path for a 1-D non-smooth length."""
    import ctypes
    from pyfft_amd import _native as N
    from pyfft_amd.generic import _chirp
    cd = numpy.dtype(dtype)
    prec = N.F64 if cd == numpy.complex128 else N.F32
    mm = ctypes.c_int32(0)
    if N.lib.mifft_bluestein_padded(prec, n, ctypes.byref(mm)) != 0:
        assert 2 * n - 1 > (5000 if prec == N.F64 else 10000)        # (round 4: padded rows up to the whole LDS of a CU)
        return
    m = mm.value
    # a smooth padded length: inside one 64 KiB tile the mixed-radix row kernel takes it too; beyond (round 4) only Bluestein does
    assert m >= 2 * n - 1 and (N.lib.mifft_mixed_supported(prec, m) == 0 or m > (2048 if prec == N.F64 else 4096))
    eps, mx = (1e-11, 1e-10) if prec == N.F64 else (1.1e-6, 1e-5)
    rows, pad = 29, 3
    rng = numpy.random.default_rng(n)
    x = (rng.standard_normal((rows, n + pad)) + 1j * rng.standard_normal((rows, n + pad))).astype(cd)
    c = _chirp(n, cd)
    b = numpy.zeros(m, numpy.complex128)
    b[:n] = numpy.conj(c)
    b[m - n + 1:] = numpy.conj(c[1:][::-1])
    tw = ctx.toGpu(numpy.exp(-2j * numpy.pi * numpy.arange(m) / m).astype(cd))
    chirp, bhat = ctx.toGpu(c.astype(cd)), ctx.toGpu((numpy.fft.fft(b) / m).astype(cd))
    a, o = ctx.toGpu(x), ctx.allocate((rows, n), cd)
    N.check(N.lib.mifft_launch_bluestein_rows(prec, n, m, rows, n + pad, n, a.ptr, o.ptr, tw.ptr, chirp.ptr, bhat.ptr, 0, 2.0, None), "blue")
    N.check(N.lib.mifft_device_sync(), "sync")
    ref = 2.0 * numpy.fft.fft(x[:, :n].astype(numpy.complex128), axis=1)
    got = o.get()
    assert numpy.abs(got - ref).sum() / numpy.abs(ref).sum() < eps
    assert numpy.abs(got - ref).max() <= mx * numpy.abs(ref).max()
    assert numpy.array_equal(a.get(), x)
    N.check(N.lib.mifft_launch_bluestein_rows(prec, n, m, rows, n, n, o.ptr, o.ptr, tw.ptr, chirp.ptr, bhat.ptr, 1, 0.5 / n, None), "blue")
    N.check(N.lib.mifft_device_sync(), "sync")
    assert numpy.abs(o.get() - x[:, :n]).sum() / numpy.abs(x[:, :n]).sum() < 2 * eps
    smooth = N.lib.mifft_mixed_supported(prec, n) == 0
    if not smooth:
        plan = ctx.getPlan((n,), dtype=dtype, any_size=True)
        assert plan._direct_blue and plan._axes[0].blue[0] == m
        y = numpy.ascontiguousarray(x[:, :n])
        d = ctx.toGpu(y)
        plan.execute(d, batch=rows)
        assert numpy.abs(d.get() - ref / 2.0).sum() / numpy.abs(ref).sum() * 2.0 < eps
        plan.execute(d, batch=rows, inverse=True)
        assert numpy.abs(d.get() - y).sum() / numpy.abs(y).sum() < 2 * eps
        assert plan._work is None


def _random_lengths(seed, count):
    rng = numpy.random.default_rng(seed)
    lens = set()
    while len(lens) < count:
        kind = rng.integers(0, 4)
        if kind == 0:
            n = int(rng.integers(2, 5000))                                   # anything: mostly Bluestein
        elif kind == 1:
            n = int(2 ** rng.integers(0, 5) * 3 ** rng.integers(0, 4) * 5 ** rng.integers(0, 3) * 7 ** rng.integers(0, 3))   # smooth
        elif kind == 2:
            n = int(rng.integers(4097, 70000))                               # beyond one tile: long smooth or multi-launch Bluestein
        else:
            n = int(rng.integers(2, 130)) * int(rng.integers(2, 130))        # composite, often smooth-ish
        if 2 <= n <= 70000:
            lens.add(n)
    return sorted(lens)


@pytest.mark.parametrize("dtype", [numpy.complex64, numpy.complex128], ids=["c64", "c128"])
def test_any_size_random_lengths(ctx, dtype):
    """A seeded sample of 1-D lengths through Plan(any_size=True) -- whichever form each one takes (dense plan, mixed-radix rows,
    two-launch long smooth, one-launch or multi-launch Bluestein) -- against numpy with the reference's thresholds: forward out of
    place with a ragged batch, inverse in place."""
    double = numpy.dtype(dtype) == numpy.complex128
    eps, mx = (1e-11, 1e-10) if double else (1.1e-6, 1e-5)
    forms = {}
    for n in _random_lengths(20261003 + double, 48):
        batch = 1 + n % 5
        rng = numpy.random.default_rng(n)
        x = (rng.standard_normal((batch, n)) + 1j * rng.standard_normal((batch, n))).astype(dtype)
        plan = ctx.getPlan((n,), dtype=dtype, any_size=True)
        form = ("dense" if not hasattr(plan, "_direct_long") else "long" if plan._direct_long is not None else "blue1" if plan._direct_blue
                else "mixed" if plan._direct_mixed else "bluestein")
        forms[form] = forms.get(form, 0) + 1
        a, b = ctx.toGpu(x), ctx.allocate(x.shape, dtype)
        plan.execute(a, b, batch=batch)
        ref = numpy.fft.fft(x.astype(numpy.complex128), axis=1)
        got = b.get()
        assert numpy.abs(got - ref).sum() / numpy.abs(ref).sum() < eps, (n, form)
        assert numpy.abs(got - ref).max() <= mx * numpy.abs(ref).max(), (n, form)
        assert numpy.array_equal(a.get(), x), (n, form)
        plan.execute(b, batch=batch, inverse=True)
        assert numpy.abs(b.get() - x).sum() / numpy.abs(x).sum() < 2 * eps, (n, form)
    assert len(forms) >= 4, forms            # the sample reaches the different forms


def test_generic_plan_reports_inner_plan_errors(ctx):
    """GenericFFTPlan.finish()/check() drain the mailboxes of the inner power-of-two plans (a persistent kernel inside a tiled
    or Bluestein plan posts its dependency time-outs there)."""
    hip = ctx.hip
    # (a tile shape WITHOUT a one-launch kernel: gather, one inner N-D plan, scatter -- since round 4 a plan that runs the tile
    # kernel builds no inner plan at all)
    plan = ctx.getPlan((16, 4), parent_shape=(64, 64), dtype=numpy.complex64, wait_for_finish=True)
    data = oracle.get_test_data((64, 64), numpy.complex64, 2, 5)
    a = ctx.toGpu(data)
    plan.execute(a, batch=2)                                  # fine
    inner = plan._inner_plans()
    assert inner
    inner[0]._mailbox = hip.ErrorMailbox()
    inner[0]._mailbox._stashed.append(("fused2", 1))          # what a timed-out persistent launch leaves behind
    with pytest.raises(RuntimeError, match="time-out"):
        plan.finish()
    plan.finish()                                             # reported once
    inner[0]._mailbox._stashed.append(("fused2", 1))
    with pytest.raises(RuntimeError, match="time-out"):
        plan.execute(a, batch=2)                              # a waiting execute() cannot return success either


def test_generic_plans_build_only_what_they_run(ctx):
    """a tiled-batch plan with a one-launch kernel and an all-smooth N-D plan hold no inner power-of-two plans
    (nothing to allocate, nothing for finish() / check() to walk); the work-array paths still build theirs."""
    tiled = ctx.getPlan((16, 16), parent_shape=(64, 64), dtype=numpy.complex64)
    assert tiled._tiled and tiled._inner_plans() == [] and tiled._tiled_tables[0] and tiled._tiled_tables[2] is None
    nd = ctx.getPlan((60, 16), dtype=numpy.complex64, any_size=True)
    assert nd._direct_nd is not None and nd._inner_plans() == [] and nd._rowplans == {}
    assert ctx.getPlan((16, 16), parent_shape=(64, 64), dtype=numpy.float32)._tiled     # (split planes: one launch too, second batch of round 4)
    work = ctx.getPlan((16, 4), parent_shape=(64, 64), dtype=numpy.complex64)           # a tile shape without a one-launch kernel: gather / N-D plan / scatter
    assert not work._tiled and len(work._inner_plans()) == 1
    blue = ctx.getPlan((4099, 4), dtype=numpy.complex64, any_size=True)                 # a long prime axis: padded power-of-two rows
    assert len(blue._inner_plans()) >= 1
    for plan, shape, batch in ((tiled, (64, 64), 2), (nd, (60, 16), 3)):
        data = _test_data(shape, numpy.complex64, batch, 96)
        a = ctx.toGpu(data)
        plan.execute(a, batch=batch)
        got = a.get().reshape((batch,) + shape)
        src = data.reshape((batch,) + shape).astype(numpy.complex128)
        if plan is tiled:
            ref = numpy.empty_like(src)
            for i in range(4):
                for j in range(4):
                    ref[:, 16 * i:16 * i + 16, 16 * j:16 * j + 16] = numpy.fft.fft2(src[:, 16 * i:16 * i + 16, 16 * j:16 * j + 16])
        else:
            ref = numpy.fft.fft2(src)
        assert numpy.abs(ref - got).sum() / numpy.abs(ref).sum() < 1.1e-6


# ---- f4 tails: smooth N-D shapes in one launch, Bluestein rows up to 5000 points in one launch ----------------------------------
@pytest.mark.parametrize("dtype", [numpy.complex64, numpy.complex128], ids=["c64", "c128"])
@pytest.mark.parametrize("shape,batch", [((100, 100), 5), ((60, 60), 7), ((30, 20, 10), 3), ((12, 20), 301), ((6, 10, 14), 33), ((100, 64), 2),
                                         ((70, 70), 3), ((9, 1, 25), 11), ((15, 16), 64)], ids=str)
def test_smooth_nd_single_launch(ctx, monkeypatch, shape, dtype, batch):
    """Every axis a smooth length and the transform inside one tile (csrc/fft_mixed_nd.hip; the reference's TODO.txt:8): ONE launch
    against numpy with the reference's thresholds, out of place (input untouched), the inverse in place, ragged last work-group,
    and against the round-3 form (one launch per axis) -- the same butterflies in the same order, so the same bits."""
    from test_generic_gpu import _run_generic
    N = ctx.hip.N
    prec = N.F32 if numpy.dtype(dtype) == numpy.complex64 else N.F64
    x, y, z = (tuple(reversed(shape)) + (1, 1))[:3]
    plan = ctx.getPlan(shape, dtype=dtype, any_size=True)
    one = N.lib.mifft_mixed_nd_supported(prec, x, y, z) == 0
    assert plan._direct_nd1 == one and plan._inner_plans() == []
    if shape == (100, 100):
        assert one == (prec == N.F32)                  # 10000 points: one fp32 tile (80 KB per LDS buffer), two launches in fp64
    _run_generic(ctx, shape, dtype, batch, seed=sum(shape))
    if one:
        rng = numpy.random.default_rng(5)
        full = (batch * shape[0],) + tuple(shape[1:])
        data = (rng.standard_normal(full) + 1j * rng.standard_normal(full)).astype(dtype)
        a = ctx.toGpu(data)
        plan.execute(a, batch=batch)
        monkeypatch.setenv("PYFFT_AMD_NO_MIXED_ND", "1")
        per_axis = ctx.getPlan(shape, dtype=dtype, any_size=True)
        assert not per_axis._direct_nd1 and per_axis._direct_nd is not None
        b = ctx.toGpu(data)
        per_axis.execute(b, batch=batch)
        assert numpy.array_equal(a.get(), b.get())


@pytest.mark.parametrize("dtype", [numpy.complex64, numpy.complex128], ids=["c64", "c128"])
@pytest.mark.parametrize("n,batch", [(2049, 9), (4099, 5), (5000, 3), (3001, 4), (2500, 7)], ids=str)
def test_bluestein_rows_beyond_one_small_tile(ctx, n, batch, dtype):
    """Lengths with a large prime factor whose padded rows take up to the whole LDS of a CU (n <= 5000 fp32 / 2500 fp64): ONE launch
    (round 3: five launches at 0.029 of the roofline for n = 4099); longer ones keep the composition.  numpy, reference thresholds."""
    from test_generic_gpu import _run_generic
    N = ctx.hip.N
    prec = N.F32 if numpy.dtype(dtype) == numpy.complex64 else N.F64
    m = ctypes.c_int32(0)
    one = N.lib.mifft_bluestein_padded(prec, n, ctypes.byref(m)) == 0
    assert one == (n <= (5000 if prec == N.F32 else 2500))
    plan = ctx.getPlan((n,), dtype=dtype, any_size=True)
    smooth = N.lib.mifft_mixed_supported(prec, n) == 0 or plan._direct_long is not None
    assert plan._direct_blue == (one and not smooth)
    _run_generic(ctx, (n,), dtype, batch, seed=n)


@pytest.mark.parametrize("dtype", [numpy.complex64, numpy.complex128], ids=["c64", "c128"])
@pytest.mark.parametrize("shape,batch", [((60, 60, 60), 2), ((24, 100, 100), 3), ((7, 90, 50), 5)], ids=str)
def test_smooth_3d_as_planes_and_lines(ctx, shape, dtype, batch):
    """3-D smooth shapes beyond one tile whose (y, x) planes fit one: the planes in ONE launch (they are more transforms of the
    2-D kernel), then the z lines -- two HBM round trips instead of three (round 3: one launch per axis).  numpy, reference
    thresholds, out of place with the input untouched, the normalised inverse in place."""
    from test_generic_gpu import _run_generic
    N = ctx.hip.N
    prec = N.F32 if numpy.dtype(dtype) == numpy.complex64 else N.F64
    z, y, x = shape
    plan = ctx.getPlan(shape, dtype=dtype, any_size=True)
    planes = N.lib.mifft_mixed_nd_supported(prec, x, y, z) != 0 and N.lib.mifft_mixed_nd_supported(prec, x, y, 1) == 0
    assert plan._direct_nd_planes == planes and not plan._direct_nd1 and plan._inner_plans() == []
    _run_generic(ctx, shape, dtype, batch, seed=sum(shape))


# ---- tiled batches on split-complex parents in one launch (csrc/fft_nd2t.hpp, SPLIT; the reference's TODO.txt:6-7) -------
@pytest.mark.parametrize("dtype", [numpy.float32, numpy.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("shape,parent", [((8, 8), (24, 40)), ((16, 16), (48, 80)), ((32, 32), (96, 64)), ((64, 64), (192, 128)),
                                          ((128, 128), (256, 384)), ((16, 32), (32, 96)), ((32, 64), (96, 64)), ((64, 128), (128, 384)),
                                          ((8, 8, 8), (16, 24, 8)), ((16, 16, 16), (32, 16, 48)), ((8, 16, 16), (8, 48, 32)),
                                          ((8, 32, 32), (24, 32, 64)), ((32, 32, 32), (64, 32, 96))], ids=str)
def test_tiled_batch_split_planes_single_launch(ctx, shape, parent, dtype, monkeypatch):
    """The tiles of split-complex parent arrays (re / im planes) transformed where they lie, ONE launch and no work array: the bits
    of the interleaved one-launch form on the same numbers, numpy tile by tile with the reference's thresholds, the input planes
    untouched, in place, the inverse, and the gather / dense plan / scatter form of the same plan."""
    from test_generic_gpu import _numpy_tiles
    if shape == (32, 32, 32) and numpy.dtype(dtype) == numpy.float64:
        shape, parent = (16, 32, 32), (32, 32, 96)
    batch = 3
    rd = numpy.dtype(dtype)
    cd = numpy.dtype(numpy.complex64 if rd == numpy.float32 else numpy.complex128)
    eps, mx = (1e-11, 1e-10) if rd == numpy.float64 else (1.1e-6, 1e-5)
    full = (batch * parent[0],) + tuple(parent[1:])
    rng = numpy.random.default_rng(177 + sum(parent))
    re, im = rng.standard_normal(full).astype(rd), rng.standard_normal(full).astype(rd)
    x = (re + 1j * im).astype(cd)
    ref = _numpy_tiles(x, batch, shape, parent)
    plan = ctx.getPlan(shape, dtype=dtype, parent_shape=parent)
    assert plan._tiled and plan._work is None and plan._inner_plans() == []
    a_re, a_im, b_re, b_im = ctx.toGpu(re), ctx.toGpu(im), ctx.allocate(full, rd), ctx.allocate(full, rd)
    plan.execute(a_re, a_im, b_re, b_im, batch=batch)
    got = b_re.get() + 1j * b_im.get()
    assert numpy.array_equal(a_re.get(), re) and numpy.array_equal(a_im.get(), im), "out-of-place execute modified its input"
    assert numpy.abs(got - ref).sum() / numpy.abs(ref).sum() < eps
    assert numpy.abs(got - ref).max() <= mx * numpy.abs(ref).max()
    # the interleaved one-launch form: same butterflies, same table factors
    iplan = ctx.getPlan(shape, dtype=cd, parent_shape=parent)
    assert iplan._tiled
    c, d = ctx.toGpu(x), ctx.allocate(full, cd)
    iplan.execute(c, d, batch=batch)
    assert numpy.array_equal(d.get(), got.astype(cd))
    plan.execute(a_re, a_im, batch=batch)                      # in place
    assert numpy.array_equal(a_re.get(), b_re.get()) and numpy.array_equal(a_im.get(), b_im.get())
    plan.execute(a_re, a_im, inverse=True, batch=batch)
    back = a_re.get() + 1j * a_im.get()
    assert numpy.abs(back - x).sum() / numpy.abs(x).sum() < 2 * eps
    assert plan._work is None
    monkeypatch.setenv("PYFFT_AMD_NO_TILED", "1")
    plan3 = ctx.getPlan(shape, dtype=dtype, parent_shape=parent)
    assert not plan3._tiled
    e_re, e_im = ctx.allocate(full, rd), ctx.allocate(full, rd)
    plan3.execute(ctx.toGpu(re), ctx.toGpu(im), e_re, e_im, batch=batch)
    three = e_re.get() + 1j * e_im.get()
    assert numpy.abs(three - got).sum() / numpy.abs(got).sum() < eps

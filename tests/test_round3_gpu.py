"""Round-3 GPU tests: the reference's error grid verbatim (test/test_errors.py:125-145), the real batch-sharded path with two
ranks on one GPU (gloo control plane), and the robustness fixes of the round (error mailbox ring, generic plans draining
their inner plans, a stream-following plan switching streams without a host sync, Plan(context=i))."""
import json
import os
import subprocess
import sys

import numpy
import pytest

import pyfft_oracle as oracle
from helpers import getDimensions
from test_errors_gpu import run_protocol

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- the reference's error grid, verbatim (test/test_errors.py:125-145) ------------------------------------------------
def _reference_grid():
    shapes = []
    for x in [3, 8, 9, 10, 11, 13, 20]:                       # 1D
        shapes.append((2 ** x,))
    for x in [4, 7, 8, 10]:                                   # 2D
        for y in [4, 7, 8, 10]:
            shapes.append((2 ** x, 2 ** y))
    for x in [4, 7, 10]:                                      # 3D
        for y in [4, 7, 10]:
            for z in [4, 7, 10]:
                shapes.append((2 ** x, 2 ** y, 2 ** z))
    batch_sizes = [1, 16, 128, 1024, 4096]
    buffer_size = 32                                          # MiB (test/test_errors.py default)
    cases = []
    for double in (False, True):
        dtypes = [numpy.float64, numpy.complex128] if double else [numpy.float32, numpy.complex64]
        for dtype in dtypes:
            for shape in shapes:
                for batch in batch_sizes:
                    x, y, z = getDimensions(shape)
                    if x * y * z * batch * dtype().nbytes > buffer_size * 1024 * 1024:
                        continue                              # (test_errors.py:143-145: skipped, not failed)
                    cases.append((shape, dtype, batch))
    return cases


@pytest.mark.parametrize("shape,dtype,batch", _reference_grid(),
                         ids=lambda v: numpy.dtype(v).name if isinstance(v, type) else str(v).replace(" ", ""))
def test_reference_error_grid(ctx, shape, dtype, batch):
    """7 + 16 + 27 shapes x batch {1, 16, 128, 1024, 4096} x both layouts per precision under the 32 MiB cap, the six
    assertions of testErrors (test/test_errors.py:18-114) with its thresholds, plus the north star's max-norm bound."""
    x, y, z = getDimensions(shape)
    run_protocol(ctx, shape, dtype, batch, seed=4321, check_oracle=(x * y * z * batch <= (1 << 16)))


# ---- pass pairs (csrc/fft_pair.hpp): 256^3 as two launches of two passes each -----------------------------------------
@pytest.mark.parametrize("dtype", [numpy.complex128, numpy.float64, numpy.complex64], ids=lambda d: numpy.dtype(d).name)
def test_pass_pairs_256_cubed(ctx, dtype):
    """BASELINE config 4's shape through the pair kernels (interleaved and split fp64, interleaved fp32): the reference's
    six-assertion protocol at batch 1 and 3, and the same data through the one-pass-per-axis chain (pairs switched off) within
    the same thresholds -- two factorisations of one transform (pyfft/kernel.py:259-283 splits a long axis the same way)."""
    from pyfft_amd import _native as N
    from pyfft_amd import passes as P
    shape = (256, 256, 256)
    plan = ctx.getPlan(shape, dtype=dtype, context=ctx.context)
    assert [k.pair_with_next for k in plan.pass_list()] == [True, False, True, False]
    # (batch 1 and 3 with the soak switch; tests/test_full_size_gpu.py runs the shape at BASELINE's batch 64 in both fp64 layouts)
    for batch in ((1, 3) if os.environ.get("PYFFT_AMD_SWEEP") else (1,)):
        run_protocol(ctx, shape, dtype, batch, seed=600 + batch, check_oracle=False)
    # pairs against the three-launch chain on the same buffer
    dt = numpy.dtype(dtype)
    split = dt.kind == "f"
    cdt = numpy.complex128 if dt in (numpy.complex128, numpy.float64) else numpy.complex64
    eps = 1e-11 if cdt == numpy.complex128 else 1.1e-6
    rng = numpy.random.default_rng(12)
    data = (rng.standard_normal((2,) + shape) + 1j * rng.standard_normal((2,) + shape)).astype(cdt)
    results = []
    for off in (0, 1):
        N.check(N.lib.mifft_debug_set(N.DEBUG_PAIR, off), "debug_set")
        try:
            pl = ctx.getPlan(shape, dtype=dtype, context=ctx.context)
            assert len(pl.pass_list()) == (3 if off else 4)
            if split:
                re, im = ctx.toGpu(numpy.ascontiguousarray(data.real)), ctx.toGpu(numpy.ascontiguousarray(data.imag))
                pl.execute(re, im, batch=2)
                results.append(re.get().astype(numpy.complex128) + 1j * im.get())
            else:
                g = ctx.toGpu(data)
                pl.execute(g, batch=2)
                results.append(g.get().astype(numpy.complex128))
        finally:
            N.check(N.lib.mifft_debug_set(N.DEBUG_PAIR, 0), "debug_set")
    assert oracle.difference(results[1], results[0], 2) < eps


# ---- fp64 strided passes of 2048 points (csrc/fft_colx.hpp): fp64 2^21 / 2^22 in two passes ---------------------------
@pytest.mark.parametrize("shape,batch", [((1 << 21,), 3), ((1 << 22,), 2), ((2048, 16), 5), ((2048, 2048), 1), ((2048, 4, 8), 3),
                                         ((2048, 2), 9)])
def test_l2048_fp64_strided_pass(ctx, shape, batch):
    """The stage-chain strided kernel for L = 2048 in fp64 (8-column tiles): the transposing first pass with the inter-pass
    twiddle (S == 1), the plain last pass (S >= 8), strided axes of 2-D / 3-D shapes, and the 4-column fallback for tiny S --
    the reference's accuracy protocol against numpy (pyfft/kernel.mako:805-1047 semantics)."""
    from pyfft_amd import _native as N
    plan = ctx.getPlan(shape, dtype=numpy.complex128, context=ctx.context)
    if shape != (2048, 2):
        assert any(p.kind == N.PASS_COL and p.L == 2048 for p in plan.pass_list()), plan.pass_list()
    if len(shape) == 1:
        assert len(plan.pass_list()) == 2
    run_protocol(ctx, shape, numpy.complex128, batch, seed=900 + len(shape) + batch, check_oracle=False)


def test_l2048_fp64_fallback_kernel(ctx):
    """The 4-column generic tile kernel behind the fast L = 2048 fp64 kernel (variant 1 = always the generic kernel), driven
    through the C ABI directly on a [2048][M * S] matrix with S = 2: compared with numpy along axis 0."""
    import ctypes
    from pyfft_amd import _native as N
    from pyfft_amd.plan import _twiddle_table
    L, S, outer = 2048, 2, 3
    rng = numpy.random.default_rng(4)
    data = (rng.standard_normal((outer, L, S)) + 1j * rng.standard_normal((outer, L, S))).astype(numpy.complex128)
    a = ctx.toGpu(data)
    b = ctx.allocate(data.shape, data.dtype)
    tw = ctx.toGpu(_twiddle_table(L, L, 1, numpy.dtype(numpy.complex128)))
    d = N.MifftPass()
    d.kind, d.precision, d.layout, d.inverse, d.L, d.variant = N.PASS_COL, N.F64, N.INTERLEAVED, 0, L, 1
    d.M, d.S, d.outer, d.outer_stride_in, d.outer_stride_out, d.scale = 1, S, outer, L * S, L * S, 1.0
    d.tw_L = tw.ptr
    N.check(N.lib.mifft_launch_pass(ctypes.byref(d), a.ptr, None, b.ptr, None, None), "launch_pass")
    N.check(N.lib.mifft_device_sync(), "sync")
    ref = numpy.fft.fft(data, axis=1)
    got = b.get()
    assert numpy.abs(got - ref).sum() / numpy.abs(ref).sum() < 1e-11


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
    from test_round2_gpu import _numpy_tiles
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


# ---- short strided passes in registers (csrc/fft_colr.hpp) -------------------------------------------------------------
@pytest.mark.parametrize("dtype", [numpy.complex64, numpy.complex128], ids=["c64", "c128"])
@pytest.mark.parametrize("L,M,S,outer", [(4, 1, 64, 3), (8, 1, 2, 5), (16, 1, 256, 2), (32, 1, 4096, 1), (16, 4, 8, 3), (8, 16, 2, 2),
                                         (32, 2, 64, 2), (4, 8, 1024, 1)], ids=str)
def test_register_only_short_strided_pass(ctx, L, M, S, outer, dtype):
    """One COL pass with L <= 32 through the C ABI: the register-only kernel (library default) against the generic LDS-staged tile
    kernel (variant 1) and against the pass algebra evaluated with numpy -- out[l][q][j] = w(L*M)^(l*q) * sum_r in[r][l][j] *
    w(L)^(r*q) (pyfft/kernel.mako:805-1047) -- forward and inverse, with and without the inter-pass twiddle."""
    import ctypes
    from pyfft_amd import _native as N
    from pyfft_amd.plan import _twiddle_table
    from pyfft_amd import passes as P
    cd = numpy.dtype(dtype)
    prec = N.F64 if cd == numpy.complex128 else N.F32
    eps = 1e-12 if prec == N.F64 else 2e-6
    rng = numpy.random.default_rng(L * 131 + M * 17 + S)
    data = (rng.standard_normal((outer, L, M, S)) + 1j * rng.standard_normal((outer, L, M, S))).astype(cd)
    a = ctx.toGpu(data)
    tw = ctx.toGpu(_twiddle_table(L, L, 1, cd))
    n = L * M
    shift = (P.log2(n) + 1) // 2
    lo = ctx.toGpu(_twiddle_table(n, 1 << shift, 1, cd))
    hi = ctx.toGpu(_twiddle_table(n, n >> shift, 1 << shift, cd))
    for inverse in (0, 1):
        x = data.astype(numpy.complex128)
        if inverse:
            x = numpy.conj(x)
        y = numpy.fft.fft(x, axis=1)                                       # [outer][q][l][j]
        ll, qq = numpy.arange(M)[None, :, None], numpy.arange(L)[:, None, None]
        y = y * numpy.exp(-2j * numpy.pi * (ll * qq) / n)[None]
        ref = numpy.transpose(y, (0, 2, 1, 3)) * 0.5                        # out[o][l][q][j], scale 0.5
        if inverse:
            ref = numpy.conj(ref)
        outs = []
        for variant in (0, 1):
            b = ctx.allocate(data.shape, cd)
            d = N.MifftPass()
            d.kind, d.precision, d.layout, d.inverse, d.L, d.variant = N.PASS_COL, prec, N.INTERLEAVED, inverse, L, variant
            d.M, d.S, d.outer, d.outer_stride_in, d.outer_stride_out, d.scale = M, S, outer, L * M * S, L * M * S, 0.5
            d.tw_L = tw.ptr
            if M > 1:
                d.tw_lo, d.tw_hi, d.tw_shift = lo.ptr, hi.ptr, shift
            N.check(N.lib.mifft_launch_pass(ctypes.byref(d), a.ptr, None, b.ptr, None, None), "launch_pass")
            N.check(N.lib.mifft_device_sync(), "sync")
            outs.append(b.get().reshape(outer, M, L, S).astype(numpy.complex128))
        for got in outs:
            assert numpy.abs(got - ref).sum() / numpy.abs(ref).sum() < eps
        assert numpy.abs(outs[0] - outs[1]).sum() / numpy.abs(ref).sum() < eps


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


WT_CASES = [((16,), 4096), ((16, 16), 512), ((64,), 999), ((1024,), 33), ((8192,), 5), ((256, 256), 3), ((1024, 1024), 2),
            ((32, 32, 128), 3), ((16, 16, 16), 37), ((128, 128, 128), 1), ((1 << 21,), 1), ((4, 2048), 9), ((512, 8), 7)]


@pytest.mark.parametrize("dtype", [numpy.complex64, numpy.complex128, numpy.float32], ids=["c64", "c128", "f32"])
@pytest.mark.parametrize("shape,batch", WT_CASES, ids=[str(c[0]) for c in WT_CASES])
def test_small_launch_write_through_is_bit_identical(ctx, shape, batch, dtype, monkeypatch):
    """Executes of <= 128 MiB per side store write-through (MIFFT_FLAG_WRITE_THROUGH on every pass): every kernel family's
    write-through form against its plain stores, bit for bit, out of place and in place, forward and inverse."""
    from pyfft_amd import _native as N
    split = numpy.dtype(dtype).kind == "f"
    cd = numpy.complex128 if dtype == numpy.complex128 else numpy.complex64
    data = oracle.get_test_data(shape, cd, batch, 77)

    def run(hints):
        if hints:
            monkeypatch.delenv("PYFFT_AMD_NO_STREAM_HINTS", raising=False)
        else:
            monkeypatch.setenv("PYFFT_AMD_NO_STREAM_HINTS", "1")
        plan = ctx.getPlan(shape, dtype=dtype, wait_for_finish=True)
        flags = [d.flags for d in plan._descriptors(batch, False, False)]
        outs = []
        for inverse in (False, True):
            if split:
                a = [ctx.toGpu(numpy.ascontiguousarray(data.real)), ctx.toGpu(numpy.ascontiguousarray(data.imag))]
                b = [ctx.allocate(data.shape, dtype), ctx.allocate(data.shape, dtype)]
                plan.execute(a[0], a[1], b[0], b[1], batch=batch, inverse=inverse)
                plan.execute(a[0], a[1], batch=batch, inverse=inverse)
                outs += [b[0].get(), b[1].get(), a[0].get(), a[1].get()]
            else:
                a, b = ctx.toGpu(data), ctx.allocate(data.shape, cd)
                plan.execute(a, b, batch=batch, inverse=inverse)
                plan.execute(a, batch=batch, inverse=inverse)
                outs += [b.get(), a.get()]
        return flags, outs

    f1, o1 = run(True)
    f0, o0 = run(False)
    assert all(f & N.FLAG_WRITE_THROUGH for f in f1) and not any(f & N.FLAG_WRITE_THROUGH for f in f0)
    for x, y in zip(o1, o0):
        assert numpy.array_equal(x, y)


def test_write_through_rule_stops_at_128_mib(ctx):
    from pyfft_amd import _native as N
    plan = ctx.getPlan((1024,), dtype=numpy.complex64)
    assert all(d.flags & N.FLAG_WRITE_THROUGH for d in plan._descriptors(16384, False, False))          # 128 MiB
    assert not any(d.flags & N.FLAG_WRITE_THROUGH for d in plan._descriptors(16385, False, False))


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


# ---- the sharded path with more than one rank --------------------------------------------------------------------------
@pytest.mark.skipif(not os.environ.get("PYFFT_AMD_SWEEP"), reason="superseded by tests/test_round5_gpu.py::test_eight_ranks_share_one_gpu (the same path with eight ranks); runs with the soak switch")
def test_two_ranks_share_one_gpu_sharded_path(tmp_path):
    """`bench.py --gpus 2` for real: two processes (one plan, stream and scratch each) on ONE device, gloo as the control
    plane, data taken from the GLOBAL dataset by transform index.  Each rank parity-checks its slice [start, start + count)
    in-process; here the first and last transform of every slice are checked again, against numpy on the global dataset
    regenerated independently of any rank."""
    sys.path.insert(0, ROOT)
    import bench
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    batch = 96                                                # per rank: 2 x 96 transforms of 8 MiB
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--control", "gloo", "--share-gpu",
                          "--config", "c2", "--batch", str(batch), "--steps", "2", "--warmup", "1", "--plain",
                          "--dump-dir", str(tmp_path)],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["config"]["global_batch"] == 2 * batch and res["scaling"] == "weak"
    ranks = res["config"]["ranks"]
    assert [r["first_transform"] for r in ranks] == [0, batch] and [r["count"] for r in ranks] == [batch, batch]
    assert all(r["parity_ok"] for r in ranks) and all(r["device"] == 0 for r in ranks)
    assert res["config"]["strategy"] == "fused2"              # the persistent kernel of the headline path, in both processes
    shape, dtname, _, seed = bench.CONFIGS["c2"]
    for g in (0, batch - 1, batch, 2 * batch - 1):
        got = numpy.load(os.path.join(str(tmp_path), "xform_%d.npy" % g))
        ref = numpy.fft.fft(bench.global_item(shape, dtname, batch, seed, g).astype(numpy.complex128))
        assert numpy.abs(got - ref).sum() / numpy.abs(ref).sum() < 1.1e-6, g
        assert numpy.abs(got - ref).max() <= 1e-5 * numpy.abs(ref).max(), g
    # weak scaling arithmetic of the line: value = all ranks' transforms over the max-over-ranks time
    assert abs(res["transforms_per_s"] - 2 * batch * res["steps"] / (res["ms_per_step"] * 1e-3 * res["steps"])) < 1e-6 * res["transforms_per_s"]


# ---- f2: the vendor comparator as a value cross-check, and the published-table benchmark ------------------------------
def _run_tool(args, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    return subprocess.run([sys.executable] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def test_hipfft_and_libmifft_agree_on_the_comparator_shapes():
    """tools/hipfft_check.py (cuda/test.cu:13-95 counterpart): hipFFT and libmifft transform the same seeded device buffer;
    values agree within the reference's thresholds on the eight comparator shapes, and both agree with numpy."""
    out = _run_tool([os.path.join(ROOT, "tools", "hipfft_check.py")])
    if out.returncode == 2:
        pytest.skip("no hipFFT library on this box")
    assert out.returncode == 0, (out.stdout[-3000:], out.stderr[-2000:])
    lines = [l for l in out.stdout.splitlines() if "L1-rel" in l]
    assert len(lines) == 8 and all(l.rstrip().endswith("ok") for l in lines), out.stdout


def test_perf_table_quick_uses_the_reference_formula():
    """tools/perf_table.py --quick (test/test_performance.py:11,22-30): batch fills the 32 MiB buffer, GFLOPS =
    5e-9 * sum(log2 dims) * points * batch / t, and the numbers are in a sane range for this part."""
    out = _run_tool([os.path.join(ROOT, "tools", "perf_table.py"), "--quick"])
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-2000:])
    rows = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert [tuple(r["shape"]) for r in rows] == [(1024,), (128, 128)]
    for r in rows:
        size = int(numpy.prod(r["shape"]))
        assert r["batch"] == (32 << 20) // (size * 8)
        want = 5.0e-9 * sum(numpy.log2(s) for s in r["shape"]) * size * r["batch"] / r["seconds_per_execute"]
        assert abs(r["gflops"] - want) < 1e-6 * want
        # one HBM round trip of a 32 MiB buffer: between 2 % and 100 % of the 8 TB/s roofline
        frac = 2.0 * size * 8 * r["batch"] / r["seconds_per_execute"] / 8e12
        assert 0.02 < frac < 1.0, frac


# ---- robustness --------------------------------------------------------------------------------------------------------
def test_error_mailbox_keeps_the_oldest_word_when_the_ring_is_full(ctx):
    """ErrorMailbox.post with SLOTS launches pending: the oldest entry is retired (its word read and kept) before its slot is
    reused; nothing is reported twice."""
    hip = ctx.hip
    box = hip.ErrorMailbox()
    bad = ctx.toGpu(numpy.array([7], dtype=numpy.uint32))
    good = ctx.toGpu(numpy.array([0], dtype=numpy.uint32))
    stream = hip.Stream()
    box.post(bad.ptr, stream, "first")
    for i in range(box.SLOTS + 5):
        box.post(good.ptr, stream, "later%d" % i)
    assert len(box._pending) <= box.SLOTS
    errors = box.collect(True)
    assert errors == [("first", 7)]
    assert box.collect(True) == [] and box._pending == []
    box.post(bad.ptr, stream, "again")
    assert box.collect(True) == [("again", 7)]


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


def test_plan_following_two_torch_streams_without_host_sync(ctx):
    """A plan built without stream= runs each execute() on torch's CURRENT stream.  Alternating two torch streams with no
    host synchronisation in between must not let the second launch's counter reset / ring writes race with the first
    persistent kernel: the plan orders the new stream behind the old one on its scratch (Context.order_scratch)."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("torch sees no GPU")
    n, batch = 1 << 18, 160                                   # fused2 (persistent, ring + counters owned by the plan; > 256 MiB per side)
    plan = ctx.getPlan((n,), dtype=numpy.complex64, wait_for_finish=False)
    assert plan.strategy(batch)[0] == "fused2"
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    g = torch.Generator(device="cpu").manual_seed(3)
    host = torch.randn(2, n, dtype=torch.complex64, generator=g)
    xs = [host[i].repeat(batch, 1).cuda() for i in range(2)]
    ys = [torch.empty_like(x) for x in xs]
    torch.cuda.synchronize()
    for rep in range(6):
        for i, s in enumerate((s1, s2)):
            with torch.cuda.stream(s):
                ret = plan.execute(xs[i], ys[i], batch=batch)
                assert int(ret.cuda_stream) == int(s.cuda_stream)
    torch.cuda.synchronize()
    plan.finish()
    for i in range(2):
        ref = numpy.fft.fft(host[i].numpy().astype(numpy.complex128))
        got = ys[i].cpu().numpy()
        for b in (0, 1, batch // 2, batch - 1):
            assert numpy.abs(got[b] - ref).sum() / numpy.abs(ref).sum() < 1.1e-6, (i, b)


def test_plan_for_a_device_given_by_index(ctx):
    """Plan(context=i) (cuda.py:121-128: the plan is built on whatever context it is given): the plan makes device i current
    around its own calls and restores the caller's.  With one visible device this exercises the guard with i == current; with
    more, a plan per device is driven from one process without the caller switching devices."""
    hip = ctx.hip
    N = hip.N
    import ctypes
    ndev = hip.device_count()
    with pytest.raises(ValueError):
        ctx.getPlan((1024,), dtype=numpy.complex64, context=ndev)      # not a visible device
    data = oracle.get_test_data((4096,), numpy.complex64, 4, 9)
    ref = oracle.numpy_fft(numpy.fft.fftn, data, 4)
    cur = ctypes.c_int()
    N.check(N.lib.mifft_get_device(ctypes.byref(cur)), "get")
    home = cur.value
    for dev in range(ndev):
        plan = ctx.getPlan((4096,), dtype=numpy.complex64, context=dev)
        assert plan._context.device == dev and plan._context._guard
        N.check(N.lib.mifft_get_device(ctypes.byref(cur)), "get")
        assert cur.value == home                               # construction restored the caller's device
        N.check(N.lib.mifft_set_device(dev), "set")            # buffers live on the plan's device
        a = ctx.toGpu(data)
        N.check(N.lib.mifft_set_device(home), "set")
        plan.execute(a, batch=4)                               # called with `home` current
        N.check(N.lib.mifft_get_device(ctypes.byref(cur)), "get")
        assert cur.value == home
        N.check(N.lib.mifft_set_device(dev), "set")
        got = a.get()
        N.check(N.lib.mifft_set_device(home), "set")
        assert oracle.difference(ref, got, 4) < 1.1e-6

"""GPU tests of the strided-axis (COL) kernels: L = 2048 fp32 / 1024 fp64 on 512-thread tiles
(csrc/fft_col3.hpp), fp64 L = 2048 on the stage-chain tiles (csrc/fft_colx.hpp) with its generic fallback, and the short passes held
entirely in registers (csrc/fft_colr.hpp).  Reference shape of the work: pyfft/kernel.py:181-283, pyfft/kernel.mako:805-1047."""
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


L2048_CASES = [((1 << 21,), 3), ((1 << 22,), 2), ((2048, 16), 5), ((2048, 64), 2), ((2048, 2, 8), 3), ((2048, 4096), 1)]


@pytest.mark.parametrize("dtype", [numpy.complex64, numpy.float32], ids=["interleaved", "split"])
@pytest.mark.parametrize("shape,batch", L2048_CASES)
def test_l2048_strided_pass(ctx, shape, batch, dtype):
    """The 512-thread L = 2048 strided-axis kernel (csrc/fft_col3.hpp): N = 2^21 / 2^22 in two passes (transposing form with
    the inter-pass twiddle + plain form), strided axes of 2-D / 3-D shapes, both layouts, forward + inverse, in place and
    out of place -- the reference's accuracy protocol against numpy."""
    import test_errors_gpu
    from pyfft_amd import _native as N
    plan = ctx.getPlan(shape, dtype=dtype, context=ctx.context)
    assert any(p.kind == N.PASS_COL and p.L == 2048 for p in plan.pass_list()), plan.pass_list()
    test_errors_gpu.run_protocol(ctx, shape, dtype, batch, seed=700 + len(shape) + batch, check_oracle=False)


@pytest.mark.parametrize("dtype", [numpy.complex128, numpy.float64], ids=["interleaved", "split"])
@pytest.mark.parametrize("shape,batch", [((1 << 19,), 3), ((1 << 20,), 2), ((1024, 16), 5), ((1024, 1024), 1), ((1024, 2, 8), 3)])
def test_l1024_fp64_strided_pass(ctx, shape, batch, dtype):
    """The fp64 form of the 512-thread strided-axis kernel (csrc/fft_col3.hpp, L = 1024): fp64 N = 2^19 / 2^20 in two passes,
    strided axes of 2-D / 3-D shapes, both layouts -- the reference's accuracy protocol against numpy."""
    import test_errors_gpu
    from pyfft_amd import _native as N
    plan = ctx.getPlan(shape, dtype=dtype, context=ctx.context)
    assert any(p.kind == N.PASS_COL and p.L == 1024 for p in plan.pass_list()), plan.pass_list()
    test_errors_gpu.run_protocol(ctx, shape, dtype, batch, seed=800 + len(shape) + batch, check_oracle=False)


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


# ---- long STRIDED axes: a y / z axis too long for one strided pass behind short rows (round 6) ------------------------------------------------
# Every (length, twiddled, stride class) of the strided-pass kernels that a default plan of up to 2^24 points selects and that no other
# default test reached (tests/test_kernel_coverage.py found them: the transposing first pass of such an axis carries the inter-pass twiddle
# with S > 1 -- S < 16: columns narrower than a tile, the generic tile kernel; S >= 16: whole 16-column tiles).  Reference shape of the
# work: pyfft/kernel.py:259-283 splits a long axis, pyfft/kernel.mako:805-1047 runs it at any stride.
_C64, _F32, _C128, _F64 = numpy.complex64, numpy.float32, numpy.complex128, numpy.float64
LONG_STRIDED_CASES = [((2, 4096, 4), _C128), ((2, 16384, 2), _F32), ((2097152,), _F64), ((2097152, 2), _C128), ((8192, 2), _F64), ((16384, 2), _C64),
                      ((4096, 2, 4), _F32)] + \
                     [((32768, 2), d) for d in (_C64, _F32, _C128, _F64)] + [((4096, 16), d) for d in (_C64, _F32, _C128)] + \
                     [((8192, 16), d) for d in (_C64, _F32)] + [((131072, 2), d) for d in (_C64, _F32, _C128, _F64)] + \
                     [((32768, 16), d) for d in (_C64, _F32, _C128, _F64)] + [((524288, 2), d) for d in (_C64, _F32, _C128, _F64)] + \
                     [((131072, 16), d) for d in (_C64, _F32, _C128, _F64)] + [((2097152, 2), d) for d in (_C64, _F32)] + \
                     [((524288, 16), d) for d in (_C64, _F32, _C128, _F64)] + [((2097152, 8), _C128)]


@pytest.mark.parametrize("shape,dtype", LONG_STRIDED_CASES, ids=lambda v: getattr(v, "__name__", str(v)).replace(" ", ""))
def test_long_strided_axes(ctx, shape, dtype):
    """The reference's six-assertion protocol against numpy (test/test_errors.py:18-114: out of place, in place, forward, inverse, input
    untouched) on shapes whose slow axis needs two strided passes; two transforms where they are small, one where one is >= 64 MiB."""
    from pyfft_amd import _native as N
    plan = ctx.getPlan(shape, dtype=dtype, context=ctx.context)
    assert sum(1 for p in plan.pass_list() if p.kind == N.PASS_COL and p.M > 1 and (p.S > 1 or len(shape) == 1)) >= 1, plan.pass_list()
    nbytes = int(numpy.prod(shape)) * (8 if numpy.dtype(dtype) in (numpy.dtype(_C64), numpy.dtype(_F32)) else 16)
    run_protocol(ctx, shape, dtype, 1 if nbytes >= (64 << 20) else 2, seed=6200 + len(shape), check_oracle=False)

"""GPU tests of the contiguous-axis kernels: the wave-autonomous small-N kernels (csrc/fft_wave.hpp) against
the LDS kernels, and the write-through store forms of every single-launch kernel family (csrc/fft_row2.hpp, fft_tile.hpp, fft_nd2.hpp).
The register-edged rows themselves: tests/test_errors_gpu.py::test_register_edged_rows.  All through the C ABI, against numpy.fft on the
complex128-upcast input (the reference's own oracle, test/test_errors.py:5-16,35) with the reference's thresholds."""
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


@pytest.mark.parametrize("dtype", [numpy.complex64, numpy.complex128, numpy.float32, numpy.float64],
                         ids=["c64", "c128", "f32", "f64"])
@pytest.mark.parametrize("n", [2, 4, 8, 16, 32, 64])
def test_small_n_wave_path_parity(ctx, n, dtype):
    """f3: N <= 32 (fp32) / <= 16 (fp64) interleaved rows run in the wave-autonomous kernel (csrc/fft_wave.hpp: no LDS,
    DPP exchange), the rest of N <= 64 and the split layouts in the LDS kernels: all against numpy, ragged batches,
    forward + inverse, out of place and in place (reference: pyfft/kernel_helpers.py:31-35, several transforms per group)."""
    import test_errors_gpu
    for batch in (1, 3, 63, 64, 65, 1000):
        test_errors_gpu.run_protocol(ctx, (n,), dtype, batch, seed=300 + n + batch, check_oracle=(batch <= 3))


def test_wave_kernel_matches_lds_kernel(ctx):
    """The wave-autonomous kernels and the LDS-staged kernels they replace give the same values (same butterflies)."""
    from pyfft_amd import _native as N
    rng = numpy.random.default_rng(11)
    for shape, batch, dtype in (((16,), 777, numpy.complex64), ((32,), 130, numpy.complex64), ((8,), 99, numpy.complex128),
                                ((16,), 65, numpy.complex128), ((16, 16), 37, numpy.complex64)):
        size = int(numpy.prod(shape))
        fdt = numpy.float32 if dtype == numpy.complex64 else numpy.float64
        data = (rng.standard_normal(size * batch).astype(fdt) + 1j * rng.standard_normal(size * batch).astype(fdt)).astype(dtype)
        outs = []
        for off in (0, 1):
            N.check(N.lib.mifft_debug_set(N.DEBUG_NO_WAVE, off), "debug_set")
            N.check(N.lib.mifft_debug_set(N.DEBUG_FORCE_WAVE, 1 - off), "debug_set")
            try:
                plan = ctx.getPlan(shape, dtype=dtype, context=ctx.context)
                a, b = ctx.toGpu(data), ctx.allocate(data.shape, data.dtype)
                plan.execute(a, b, batch=batch)
                outs.append(b.get())
            finally:
                N.check(N.lib.mifft_debug_set(N.DEBUG_NO_WAVE, 0), "debug_set")
                N.check(N.lib.mifft_debug_set(N.DEBUG_FORCE_WAVE, 0), "debug_set")
        ref = numpy.fft.fftn(data.astype(numpy.complex128).reshape((batch,) + shape), axes=tuple(range(1, len(shape) + 1))).ravel()
        for o in outs:
            assert numpy.abs(o - ref).max() <= (1e-5 if dtype == numpy.complex64 else 1e-10) * numpy.abs(ref).max()
        assert numpy.allclose(outs[0], outs[1], rtol=0, atol=(2e-6 if dtype == numpy.complex64 else 1e-14) * numpy.abs(ref).max())


@pytest.mark.parametrize("batch", [1, 7, 8, 9, 100, 1025])
def test_16x16_wave_plane_kernel(ctx, batch):
    """The (16, 16) fp32 plane in the wave-autonomous kernel (it runs on request only: the LDS kernel is faster)."""
    import test_errors_gpu
    from pyfft_amd import _native as N
    N.check(N.lib.mifft_debug_set(N.DEBUG_FORCE_WAVE, 1), "debug_set")
    try:
        test_errors_gpu.run_protocol(ctx, (16, 16), numpy.complex64, batch, seed=500 + batch, check_oracle=(batch <= 9))
    finally:
        N.check(N.lib.mifft_debug_set(N.DEBUG_FORCE_WAVE, 0), "debug_set")


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

"""GPU tests of the pass-pair kernels (csrc/fft_pair.hpp, fft_pair_f32.hip, fft_pair_f64.hip): two consecutive passes of a chain in one launch
-- 256^3 as two launches of two passes each, a 4096-point y axis as pair + pass, short (y, z) behind a long x as row + pair.  Every pair
instance a default plan selects is a case here (tests/test_kernel_coverage.py enforces it).  The reference factors a long axis the same way:
pyfft/kernel.py:259-283."""
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


def test_pass_pairs_256_cubed_alternative_split(ctx):
    """The other y split of 256^3 in complex128 (y = 64 x 4 instead of 32 x 8; MIFFT_DEBUG_PAIR = 2, the A/B form of tools/pair_probe.py,
    profiles/r03_b_c4_pair_split.log) is a kernel instance of the library that no default plan selects: the six-assertion protocol, so that
    it has a test of its own."""
    from pyfft_amd import _native as N
    N.check(N.lib.mifft_debug_set(N.DEBUG_PAIR, 2), "debug_set")
    try:
        plan = ctx.getPlan((256, 256, 256), dtype=numpy.complex128)
        assert [k.pair_with_next for k in plan.pass_list()] == [True, False, True, False] and plan.pass_list()[1].L == 64
        run_protocol(ctx, (256, 256, 256), numpy.complex128, 1, seed=660, check_oracle=False)
    finally:
        N.check(N.lib.mifft_debug_set(N.DEBUG_PAIR, 0), "debug_set")


# ---- one pass pair instead of a third launch (csrc/fft_pair_f32.hip / _f64.hip, pyfft_amd/passes.py) ---------------------------------
PAIR_CHAIN_CASES = [((4096, 256), numpy.complex64, 2), ((4096, 256), numpy.complex128, 1),
                    ((32, 32, 2048), numpy.complex64, 2), ((16, 16, 2048), numpy.complex64, 5),
                    ((32, 32, 1024), numpy.complex128, 2), ((16, 16, 1024), numpy.complex128, 5), ((2, 4096, 256), numpy.complex64, 1)]


_PAIR_CHAIN_SOAK = [((4096, 512), numpy.complex64, 1), ((4096, 512), numpy.complex128, 1), ((32, 32, 4096), numpy.complex64, 1),
                    ((16, 16, 4096), numpy.complex64, 1), ((32, 32, 2048), numpy.complex128, 1), ((16, 16, 2048), numpy.complex128, 2)]


PAIR_CHAIN_CASES += [((4096, 128), numpy.complex128, 3), ((4096, 128), numpy.complex64, 5)]      # (late in round 5: 128-point rows)
# (round 6, found by tests/test_kernel_coverage.py: the YZ kernel keyed (S0 = 16384, R1 = 4, nz = 256) -- built for the second pair of 256^3
# in the split layout -- is also what an interleaved (256, 4, 16384) plan selects for its y and z axes: 256 MiB per transform)
PAIR_CHAIN_CASES += [((256, 4, 16384), numpy.complex128, 1)]


# every one of these is a pair-kernel instance of its own (csrc/fft_pair_f32.hip / _f64.hip) that a default plan selects: all in the default suite
# (round 6; the biggest planes are 16 ... 256 MiB per transform: numpy takes a second or two per case)
PAIR_CHAIN_CASES += _PAIR_CHAIN_SOAK + [((4096, 4096), numpy.complex128, 1),
                                        ((4096, 1024), numpy.complex64, 3), ((4096, 2048), numpy.complex64, 1), ((4096, 4096), numpy.complex64, 1),
                                        ((4096, 1024), numpy.complex128, 2), ((4096, 2048), numpy.complex128, 1)]


@pytest.mark.parametrize("shape,dtype,batch", PAIR_CHAIN_CASES, ids=lambda v: getattr(v, "__name__", str(v)).replace(" ", ""))
def test_pair_chains_two_launches(ctx, shape, dtype, batch):
    """Shapes that ran THREE launches through round 4 -- a 2-D shape with a 4096-point y axis (row + two strided passes: the reference's
    own factorisation of a long axis, pyfft/kernel.py:259-283) and a 3-D shape with short y and z behind a long x (one chain per axis,
    pyfft/plan.py:160-167) -- now run one pass PAIR and one single pass.  The reference's six-assertion protocol against numpy
    (test/test_errors.py:18-114: out of place, in place, forward, inverse, input untouched) at ragged batches, and the chain really has
    two launches."""
    from test_errors_gpu import run_protocol
    from pyfft_amd.passes import launch_units
    plan = ctx.getPlan(shape, dtype=dtype)
    units = launch_units(plan.pass_list())
    assert len(units) == (3 if len(shape) == 3 and shape[0] == 2 else 2), plan.pass_list()
    assert sum(1 for k in plan.pass_list() if k.pair_with_next) == 1
    run_protocol(ctx, shape, dtype, batch, seed=5150 + batch)


@pytest.mark.parametrize("shape,dtype,batch", [((4096, 256), numpy.complex64, 320), ((32, 32, 2048), numpy.complex64, 160), ((4096, 512), numpy.complex128, 96)],
                         ids=lambda v: getattr(v, "__name__", str(v)).replace(" ", ""))
def test_pair_chains_pipelined_chunks(ctx, shape, dtype, batch):
    """The same chains beyond the chain threshold: cache-sized chunks on two streams, a pair launch + a single launch per chunk.
    The whole array against the plain chain's (same kernels: identical bits), sampled transforms against numpy."""
    hip = ctx.hip
    size = int(numpy.prod(shape))
    cdt = numpy.dtype(dtype)
    data = _tiled_noise(size * batch, dtype, 777)
    tol, tol_max = (1.1e-6, 1e-5) if cdt == numpy.complex64 else (1e-11, 1e-10)
    outs = {}
    for strat in ("chain", "auto"):
        os.environ["PYFFT_AMD_STRATEGY"] = strat
        try:
            plan = hip.Plan(shape, dtype=dtype)
            st = plan.strategy(batch)[0]
            assert st == ("chain" if strat == "chain" else "pipelined"), st
            a, b = hip.to_gpu(data), hip.DeviceArray((size * batch,), dtype)
            plan.execute(a, b, batch=batch)
            outs[strat] = b.get()
        finally:
            os.environ.pop("PYFFT_AMD_STRATEGY", None)
    assert numpy.array_equal(outs["chain"].view(numpy.uint8), outs["auto"].view(numpy.uint8))
    for item in (0, batch // 2, batch - 1):
        ref = numpy.fft.fftn(data[item * size:(item + 1) * size].astype(numpy.complex128).reshape(shape)).reshape(-1)
        got = outs["auto"][item * size:(item + 1) * size].astype(numpy.complex128)
        assert numpy.abs(ref - got).sum() / numpy.abs(ref).sum() < tol and numpy.abs(ref - got).max() <= tol_max * numpy.abs(ref).max()


# ---- 3-D shapes with 256-point rows next to a shorter axis: the two pass pairs of 256^3 instead of three launches ---------------------------
_LATE_PAIR_SHAPES = [((64, 128, 256), numpy.complex128, 1), ((64, 256, 256), numpy.complex64, 1), ((32, 256, 128), numpy.complex128, 3)]
# (round 6: every shape is a pair-kernel instance of its own -- all of them in the default suite)
_LATE_PAIR_SHAPES += [((64, 256, 128), numpy.complex128, 1), ((128, 256, 128), numpy.complex128, 1), ((256, 256, 128), numpy.complex128, 1),
                      ((32, 256, 256), numpy.complex128, 1), ((32, 128, 256), numpy.complex128, 2),
                      ((128, 256, 64), numpy.complex128, 1), ((32, 256, 64), numpy.complex128, 5), ((256, 256, 64), numpy.complex128, 1),
                      ((64, 256, 64), numpy.complex128, 2), ((32, 256, 256), numpy.complex64, 3),
                      ((128, 256, 256), numpy.complex128, 1), ((64, 256, 256), numpy.complex128, 1), ((128, 128, 256), numpy.complex128, 2),
                      ((256, 128, 256), numpy.complex128, 1), ((128, 256, 256), numpy.complex64, 1)]


@pytest.mark.parametrize("shape,dtype,batch", _LATE_PAIR_SHAPES, ids=lambda v: getattr(v, "__name__", str(v)).replace(" ", ""))
def test_pass_pairs_for_256_point_rows(ctx, monkeypatch, shape, dtype, batch):
    """(z, y, 256) with y in {128, 256}, z in {32 ... 256}, and (z, 256, 128) in complex128; (z, 256, 256), z in {64, 128}, in complex64: (ROW x, COL y R0) and
    (COL y R1, COL z) as two launches (csrc/fft_pair_f64.hip: y = 32 x 8 / 32 x 4; fft_pair_f32.hip: 64 x 4; pyfft/kernel.py:259-283
    splits a long axis the same way).  The reference's six-assertion protocol against numpy, and the same data through the one-pass-per-axis chain (pairs
    switched off) within the same thresholds."""
    from test_errors_gpu import run_protocol
    from pyfft_amd import _native as N
    from pyfft_amd.passes import launch_units
    plan = ctx.getPlan(shape, dtype=dtype)
    assert len(launch_units(plan.pass_list())) == 2 and sum(1 for k in plan.pass_list() if k.pair_with_next) == 2, plan.pass_list()
    run_protocol(ctx, shape, dtype, batch, seed=8100 + shape[0])
    if (shape, dtype, batch) not in _LATE_PAIR_SHAPES[:3]:
        return              # (the same data through the three-launch chain: a check of the test data, run for three of the shapes)
    N.lib.mifft_debug_set(N.DEBUG_PAIR, 1)
    try:
        chain = ctx.getPlan(shape, dtype=dtype)
        # (three launches, or two where the (y, x) plane has a one-tile kernel: (z, 256, 64) in fp64 -- the pairs measured 0.29 -> 0.39 there)
        assert len(launch_units(chain.pass_list())) in (2, 3) and not any(k.pair_with_next for k in chain.pass_list())
        run_protocol(ctx, shape, dtype, batch, seed=8100 + shape[0], check_oracle=False)
    finally:
        N.lib.mifft_debug_set(N.DEBUG_PAIR, 0)

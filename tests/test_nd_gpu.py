"""GPU tests of the one-launch N-D kernels beyond the fixed-shape table (that table: tests/test_errors_gpu.py::test_fixed_shape_nd_kernels):
several work-groups per transform (csrc/fft_nd2z.hpp) and the tiny shapes the tuning table routes to the run-time-shaped kernel
(csrc/fft_nd.hpp).  Reference: one chain per axis, pyfft/plan.py:135-171."""
import os

import numpy
import pytest

from helpers import EPS_F, MAX_F, _noise, _tiled_noise
from test_errors_gpu import run_protocol

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- one-tile-per-CU N-D shapes as two work-groups per transform (csrc/fft_nd2z.hpp) ------------------------------------------------
ND2Z_SHAPES = {numpy.complex64: [(1024, 32), (512, 64), (256, 128), (128, 256), (32, 1024), (8, 64, 64), (16, 16, 128), (32, 32, 32),
                                 # (two-per-CU shapes: split in small launches only -- 11 transforms are one)
                                 (1024, 16), (512, 32), (256, 64), (128, 128), (64, 256), (32, 512), (16, 1024), (16, 32, 32), (32, 16, 32), (32, 32, 16)],
               numpy.complex128: [(512, 32), (256, 64), (64, 256), (32, 512), (16, 32, 32), (16, 16, 64), (128, 128), (64, 16, 16),
                                  (512, 16), (256, 32), (128, 64), (64, 128), (32, 256), (16, 512), (16, 16, 32), (16, 32, 16), (32, 16, 16)]}


@pytest.mark.parametrize("dtype", [numpy.complex64, numpy.complex128], ids=["c64", "c128"])
def test_two_work_groups_per_transform_nd(ctx, dtype):
    """The N-D shapes of 32768 points (fp32) / 16384 points (fp64) as TWO work-groups per transform (one decimation-in-frequency step
    along the slowest axis folded into the loads, each half on a two-per-CU tile; numpy shapes = (z, y, x)): every such shape at a
    ragged batch against numpy with the reference's thresholds (test/test_errors.py:20-23), forward and inverse, against the
    one-tile-per-CU kernel to rounding, and an in-place call -- which must take the one-tile kernel -- bit-identical to it."""
    from pyfft_amd import _native as N
    hip = ctx.hip
    cdt = numpy.dtype(dtype)
    tol, tol_max, tol_same = (1.1e-6, 1e-5, 5e-7) if cdt == numpy.complex64 else (1e-11, 1e-10, 1e-14)
    for shape in ND2Z_SHAPES[dtype]:
        size = int(numpy.prod(shape))
        batch = 11
        data = _tiled_noise(size * batch, dtype, 4400 + shape[0])
        plan = hip.Plan(shape, dtype=dtype)
        assert len(plan.pass_list()) == 1 and plan.pass_list()[0].kind == N.PASS_ND, (shape, plan.pass_list())
        a = hip.to_gpu(data)
        outs = {}
        for alt in (6, 5):                       # 6: the one-tile-per-CU kernel, 5 (= the default): two work-groups per transform
            N.check(N.lib.mifft_debug_set(N.DEBUG_ALT_ROWS, alt), "debug_set")
            try:
                b = hip.DeviceArray((size * batch,), dtype)
                plan.execute(a, b, batch=batch)
                outs[alt] = b.get()
                if alt == 5:
                    plan.execute(b, batch=batch, inverse=True)          # in place: the one-tile kernel whatever the switch says
                    back = b.get()
                    c = hip.DeviceArray((size * batch,), dtype)
                    plan.execute(hip.to_gpu(outs[5]), c, batch=batch, inverse=True)     # out of place: the two-work-group form
                    back2 = c.get()
            finally:
                N.check(N.lib.mifft_debug_set(N.DEBUG_ALT_ROWS, 0), "debug_set")
        assert numpy.array_equal(a.get(), data), "input modified"
        for item in range(batch):
            sl = slice(item * size, (item + 1) * size)
            ref = numpy.fft.fftn(data[sl].astype(numpy.complex128).reshape(shape)).reshape(-1)
            got = outs[5][sl].astype(numpy.complex128)
            assert numpy.abs(ref - got).sum() / numpy.abs(ref).sum() < tol, (shape, item)
            assert numpy.abs(ref - got).max() <= tol_max * numpy.abs(ref).max(), (shape, item)
        d = numpy.abs(outs[5].astype(numpy.complex128) - outs[6]).sum() / numpy.abs(outs[6]).sum()
        assert d < tol_same, (shape, d)
        for inv in (back, back2):
            assert numpy.abs(inv.astype(numpy.complex128) - data).sum() / numpy.abs(data).sum() < tol, shape


OOP_ND_CASES = [(sh, numpy.complex64) for sh in [(256, 256), (512, 128), (1024, 64), (64, 1024), (16, 64, 64),
                                                  # 32768-point shapes without a one-tile kernel: two work-groups per transform
                                                  (64, 512), (16, 2048), (2048, 16), (64, 8, 64), (16, 128, 16), (128, 16, 16), (32, 16, 64), (32, 64, 16),
                                                  # (round 6: the three instances the coverage rule found without a test -- (z, y, x))
                                                  (64, 16, 32), (16, 64, 32), (16, 32, 64)]] + \
               [(sh, numpy.complex128) for sh in [(16, 1024), (1024, 16), (8, 32, 64), (64, 8, 32), (4, 64, 64), (64, 4, 64), (32, 16, 32), (16, 64, 16), (32, 32, 16),
                                                  (8, 64, 32), (32, 8, 64)]]


@pytest.mark.parametrize("shape,dtype", OOP_ND_CASES, ids=lambda v: getattr(v, "__name__", "x".join(map(str, v)) if isinstance(v, tuple) else str(v)))
def test_four_work_groups_per_transform_out_of_place(ctx, shape, dtype, monkeypatch):
    """Shapes with a one-launch kernel for OUT-OF-PLACE executes only (csrc/fft_nd2z.hpp; two launches as a chain, which the plan keeps for
    in-place executes): 65536 points (fp32) on four work-groups per transform, and the 32768-point (fp32) / 16384-point (fp64) shapes that
    have no one-tile kernel on two.  The reference's thresholds against numpy at a ragged batch, forward and inverse, out of place and in
    place, and the chain's result to rounding."""
    hip = ctx.hip
    size = int(numpy.prod(shape))
    csz = numpy.dtype(dtype).itemsize
    batch = (261 * 65536 * 8) // (size * csz)                     # 130.5 MiB per side: beyond half the cache, where the plan uses every such kernel
    EPS, MAXN, SAME = (EPS_F, MAX_F, 5e-7) if numpy.dtype(dtype) == numpy.complex64 else (1e-11, 1e-10, 1e-14)
    data = _tiled_noise(size * batch, dtype, 4500 + shape[0])
    plan = hip.Plan(shape, dtype=dtype)
    assert plan._oop_nd is not None and len(plan.pass_list()) == 2, plan.pass_list()
    assert batch * size * csz > plan._context.machine.write_through_max_bytes
    a = hip.to_gpu(data)
    b = hip.DeviceArray((size * batch,), dtype)
    plan.execute(a, b, batch=batch)
    got = b.get()
    assert numpy.array_equal(a.get(), data), "input modified"
    for item in (0, 1, 7, 8, 9, batch // 2, batch - 6, batch - 5, batch - 1):      # (groups of eight transforms share an XCD: both ends of the last, ragged group)
        sl = slice(item * size, (item + 1) * size)
        ref = numpy.fft.fftn(data[sl].astype(numpy.complex128).reshape(shape)).reshape(-1)
        g = got[sl].astype(numpy.complex128)
        assert numpy.abs(ref - g).sum() / numpy.abs(ref).sum() < EPS, (shape, item)
        assert numpy.abs(ref - g).max() <= MAXN * numpy.abs(ref).max(), (shape, item)
    c = hip.to_gpu(data)
    plan.execute(c, batch=batch)                                  # in place: the chain
    inplace = c.get()
    assert numpy.abs(inplace.astype(numpy.complex128) - got).sum() / numpy.abs(got).sum() < SAME
    monkeypatch.setenv("PYFFT_AMD_NO_OOP_ND", "1")
    d = hip.DeviceArray((size * batch,), dtype)
    hip.Plan(shape, dtype=dtype).execute(a, d, batch=batch)       # out of place on the chain
    assert numpy.array_equal(d.get(), inplace)
    monkeypatch.delenv("PYFFT_AMD_NO_OOP_ND")
    plan.execute(b, a, batch=batch, inverse=True)                 # inverse, out of place
    assert numpy.abs(a.get().astype(numpy.complex128) - data).sum() / numpy.abs(data).sum() < EPS


# ---- round 6: the same for split-complex planes (csrc/fft_nd2zp.hpp) ------------------------------------------------------------------------
OOP_ND_SPLIT_CASES = [(128, 256), (256, 128), (512, 64), (64, 512), (16, 2048), (2048, 16), (8, 64, 64), (64, 8, 64), (16, 128, 16),
                      (128, 16, 16), (16, 32, 64), (32, 16, 64), (16, 64, 32), (64, 16, 32), (32, 64, 16), (16, 16, 128),
                      # 65536 points: four quarters, in large launches
                      (256, 256), (512, 128), (1024, 64), (64, 1024), (16, 64, 64)]
OOP_ND_SPLIT_LARGE_ONLY = {(64, 512), (256, 256), (512, 128), (1024, 64), (64, 1024), (16, 64, 64)}


@pytest.mark.parametrize("shape", OOP_ND_SPLIT_CASES, ids=lambda v: "x".join(map(str, v)))
@pytest.mark.parametrize("small", [False, True], ids=["large", "small"])
def test_two_work_groups_per_transform_split_planes(ctx, shape, small, monkeypatch):
    """float32 planes (pyfft/plan.py:26-35) of the 32768- / 65536-point shapes: out of place ONE launch on two half-size (four quarter-size)
    work-groups per transform that move 16 bytes per lane and plane (csrc/fft_nd2zp.hpp); in place the plan's chain ((16, 16, 128): its one-tile kernel).  A large
    launch (130 MiB per side: non-temporal stores) and a small ragged one (write-through stores, the last group of eight transforms
    partly empty): the reference's thresholds against numpy on sampled transforms, input untouched, the in-place result and the
    interleaved twin to rounding, the route switched off (PYFFT_AMD_NO_OOP_ND / MIFFT_DEBUG_ALT_ROWS = 6), the inverse."""
    from pyfft_amd import _native as N
    hip = ctx.hip
    size = int(numpy.prod(shape))
    batch = 13 if small else (261 * 65536 * 8) // (size * 8)
    rng = numpy.random.default_rng(4700 + shape[0] + small)
    re = _noise(rng, size * batch, numpy.float32)
    im = _noise(rng, size * batch, numpy.float32)
    plan = hip.Plan(shape, dtype=numpy.float32)
    one_launch = len(plan.pass_list()) == 1
    assert one_launch == (shape == (16, 16, 128)) and (one_launch or plan._oop_nd is not None), plan.pass_list()
    # ((64, 512) loses to its two launches at 32 MiB, the 65536-point shapes like their interleaved twins: large launches only)
    assert plan._oop_any_size == (shape not in OOP_ND_SPLIT_LARGE_ONLY) or one_launch
    if not one_launch and not plan._oop_any_size and small:
        assert plan.strategy(batch, inplace=False)[0] == "chain"
        return
    assert one_launch or plan.strategy(batch, inplace=False) == ("nd_oop",), plan.strategy(batch, inplace=False)
    a_re, a_im = hip.to_gpu(re), hip.to_gpu(im)
    b_re, b_im = hip.DeviceArray((size * batch,), numpy.float32), hip.DeviceArray((size * batch,), numpy.float32)
    plan.execute(a_re, a_im, b_re, b_im, batch=batch)
    got = b_re.get().astype(numpy.complex128) + 1j * b_im.get()
    assert numpy.array_equal(a_re.get(), re) and numpy.array_equal(a_im.get(), im), "input modified"
    x = re.astype(numpy.complex128) + 1j * im
    for item in sorted({0, 1, 7, 8, 9, batch // 2, batch - 6, batch - 5, batch - 1}):
        sl = slice(item * size, (item + 1) * size)
        ref = numpy.fft.fftn(x[sl].reshape(shape)).reshape(-1)
        assert numpy.abs(ref - got[sl]).sum() / numpy.abs(ref).sum() < EPS_F, (shape, item)
        assert numpy.abs(ref - got[sl]).max() <= MAX_F * numpy.abs(ref).max(), (shape, item)
    c_re, c_im = hip.to_gpu(re), hip.to_gpu(im)
    plan.execute(c_re, c_im, batch=batch)                         # in place: the chain / the one-tile kernel
    inplace = c_re.get().astype(numpy.complex128) + 1j * c_im.get()
    assert numpy.abs(inplace - got).sum() / numpy.abs(got).sum() < 5e-7
    twin = hip.Plan(shape, dtype=numpy.complex64)
    t_in = hip.to_gpu((re + 1j * im).astype(numpy.complex64))
    t_out = hip.DeviceArray((size * batch,), numpy.complex64)
    twin.execute(t_in, t_out, batch=batch)
    assert numpy.abs(t_out.get().astype(numpy.complex128) - got).sum() / numpy.abs(got).sum() < 5e-7
    # the route switched off: the result of the in-place execute bit for bit
    if one_launch:
        N.check(N.lib.mifft_debug_set(N.DEBUG_ALT_ROWS, 6), "debug_set")
    else:
        monkeypatch.setenv("PYFFT_AMD_NO_OOP_ND", "1")
    try:
        d_re, d_im = hip.DeviceArray((size * batch,), numpy.float32), hip.DeviceArray((size * batch,), numpy.float32)
        hip.Plan(shape, dtype=numpy.float32).execute(a_re, a_im, d_re, d_im, batch=batch)
        assert numpy.array_equal(d_re.get().astype(numpy.complex128) + 1j * d_im.get(), inplace)
    finally:
        N.check(N.lib.mifft_debug_set(N.DEBUG_ALT_ROWS, 0), "debug_set")
        monkeypatch.delenv("PYFFT_AMD_NO_OOP_ND", raising=False)
    plan.execute(b_re, b_im, a_re, a_im, batch=batch, inverse=True)     # inverse, out of place
    back = a_re.get().astype(numpy.complex128) + 1j * a_im.get()
    assert numpy.abs(back - x).sum() / numpy.abs(x).sum() < EPS_F


# ---- tiny one-launch N-D shapes routed to the run-time-shaped kernel (tuning table "nd_generic") -------------------------------------------
@pytest.mark.parametrize("shape,dtype,batch", [((16, 2), numpy.complex64, 37), ((2, 8), numpy.complex64, 100), ((4, 4), numpy.complex128, 61)],
                         ids=lambda v: getattr(v, "__name__", str(v)).replace(" ", ""))
def test_tiny_nd_shapes_small_launches(ctx, monkeypatch, shape, dtype, batch):
    """Shapes of the tuning table's "nd_generic" lists in small launches (the "always" shape on the run-time-shaped kernel, the "big" ones on
    their fixed instances): the reference's six-assertion protocol against numpy (test/test_errors.py:18-114), either way round."""
    from test_errors_gpu import run_protocol
    run_protocol(ctx, shape, dtype, batch, seed=977)
    monkeypatch.setenv("PYFFT_AMD_NO_ND_GENERIC", "1")
    run_protocol(ctx, shape, dtype, batch, seed=977, check_oracle=False)


@pytest.mark.parametrize("shape,dtype,batch", [((16, 2), numpy.complex64, 700001), ((8, 8), numpy.complex64, 270001), ((4, 4), numpy.complex128, 530001),
                                               ((4, 2), numpy.complex64, 2100001), ((2, 8), numpy.complex64, 1100001), ((8, 2), numpy.complex128, 600001)],
                         ids=lambda v: getattr(v, "__name__", str(v)).replace(" ", ""))
def test_tiny_nd_shapes_big_launches(ctx, monkeypatch, shape, dtype, batch):
    """The same shapes in launches beyond the size rule (130-270 MiB per side, ragged batches), where the plan marks the pass variant 1: the
    run-time-shaped kernel's result against the fixed instance's over the WHOLE array (same transform, another operation order: the
    reference's L1 threshold and the north star's max-norm bound), 64 sampled transforms against numpy, in place == out of place, the
    input untouched, and the inverse round trip."""
    hip = ctx.hip
    size = int(numpy.prod(shape))
    cdt = numpy.dtype(dtype)
    eps, mx = (1.1e-6, 1e-5) if cdt == numpy.complex64 else (1e-11, 1e-11)
    data = _tiled_noise(size * batch, dtype, 611)
    plan = hip.Plan(shape, dtype=dtype, wait_for_finish=True)
    assert plan._descriptors(batch, False, False)[0].variant == 1
    a, b = hip.to_gpu(data), hip.DeviceArray((size * batch,), dtype)
    plan.execute(a, b, batch=batch)
    got = b.get()
    assert numpy.array_equal(a.get(), data), "an out-of-place execute touched its input"
    c = hip.to_gpu(data)
    plan.execute(c, batch=batch)
    assert numpy.array_equal(c.get(), got), "in place differs from out of place"
    plan.execute(c, batch=batch, inverse=True)
    back = c.get()
    assert numpy.abs(back - data).sum() / numpy.abs(data).sum() < eps
    for item in numpy.linspace(0, batch - 1, 64).astype(int):
        ref = numpy.fft.fftn(data[item * size:(item + 1) * size].reshape(shape).astype(numpy.complex128)).reshape(-1)
        g = got[item * size:(item + 1) * size]
        assert numpy.abs(ref - g).sum() / numpy.abs(ref).sum() < eps and numpy.abs(ref - g).max() <= max(mx, 1e-5 if cdt == numpy.complex64 else 1e-11) * numpy.abs(ref).max(), item
    monkeypatch.setenv("PYFFT_AMD_NO_ND_GENERIC", "1")
    fixed = hip.Plan(shape, dtype=dtype, wait_for_finish=True)
    assert fixed._descriptors(batch, False, False)[0].variant == 0
    fixed.execute(a, b, batch=batch)
    want = b.get()
    assert numpy.abs(want - got).sum() / numpy.abs(want).sum() < eps
    assert numpy.abs(want - got).max() <= 1e-5 * numpy.abs(want).max()


# ---- the run-time-shaped N-D kernel by tile size class, and dense split-complex planes on the tiled fixed-shape kernels (round 6) --------------
# (what tests/test_kernel_coverage.py found without a default test: shapes with two-point rows of every size up to one tile in the four
# dtypes -- no fixed instance exists for them, the ONE run-time-shaped kernel of csrc/fft_nd.hpp takes them with another tile / radix
# choice per size -- and the split-complex shapes whose dense planes run the tiled fixed-shape kernel of csrc/fft_nd2t.hpp)
GENERIC_ND_CASES = [((1 << (k - 1), 2), dt) for dt in (numpy.complex64, numpy.float32, numpy.complex128, numpy.float64) for k in range(2, 15)
                    if (1 << k) <= (16384 if dt in (numpy.complex64, numpy.float32) else 8192)] + \
                   [((1 << (k - 2), 2, 2), dt) for dt in (numpy.complex64, numpy.float32, numpy.complex128, numpy.float64) for k in range(3, 15)
                    if (1 << k) <= (16384 if dt in (numpy.complex64, numpy.float32) else 8192)] + [((4, 4096), numpy.complex64)]


@pytest.mark.parametrize("shape,dtype", GENERIC_ND_CASES, ids=lambda v: getattr(v, "__name__", str(v)).replace(" ", ""))
def test_run_time_shaped_nd_kernel_size_classes(ctx, shape, dtype):
    """One launch, any power-of-two shape inside a tile (pyfft/plan.py:135-171 runs one chain per axis): the reference's six-assertion
    protocol at a ragged batch."""
    from pyfft_amd import _native as N
    plan = ctx.getPlan(shape, dtype=dtype)
    assert len(plan.pass_list()) == 1 and plan.pass_list()[0].kind == N.PASS_ND, plan.pass_list()
    size = int(numpy.prod(shape))
    run_protocol(ctx, shape, dtype, 5 if size >= 4096 else 4099 // size, seed=6300 + size % 89, check_oracle=size <= 1024)


DENSE_PLANES_CASES = [((16, 32), numpy.float32, 37), ((32, 64), numpy.float32, 21), ((8, 32, 32), numpy.float32, 5), ((16, 32), numpy.float64, 37),
                      ((32, 32), numpy.float64, 19), ((32, 64), numpy.float64, 11), ((8, 16, 16), numpy.float64, 9), ((8, 32, 32), numpy.float64, 3),
                      ((16, 32, 32), numpy.float64, 3)]


@pytest.mark.parametrize("shape,dtype,batch", DENSE_PLANES_CASES, ids=lambda v: getattr(v, "__name__", str(v)).replace(" ", ""))
def test_dense_split_planes_on_the_tiled_fixed_kernels(ctx, shape, dtype, batch):
    """float32 / float64 plans (the reference's split layout, pyfft/plan.py:26-35) whose whole transform is one launch of the tiled
    fixed-shape kernel with one tile per "parent" (rows of >= 128 bytes per plane): the six-assertion protocol."""
    from pyfft_amd import _native as N
    plan = ctx.getPlan(shape, dtype=dtype)
    assert len(plan.pass_list()) == 1 and plan.pass_list()[0].kind == N.PASS_ND, plan.pass_list()
    run_protocol(ctx, shape, dtype, batch, seed=6400 + batch, check_oracle=False)


# ---- dense split-complex N-D shapes on 16-byte plane accesses (csrc/fft_nd2p.hpp, round 6) ------------------------------------------------
PLANES16_CASES = [(sh, numpy.float32) for sh in [(16, 16), (32, 32), (64, 64), (16, 16, 16), (8, 8, 64), (16, 16, 128), (32, 32, 32)]] + \
                 [(sh, numpy.float64) for sh in [(16, 16), (32, 32), (64, 64), (16, 16, 16), (8, 8, 64)]]


@pytest.mark.parametrize("shape,dtype", PLANES16_CASES, ids=lambda v: getattr(v, "__name__", str(v)).replace(" ", ""))
def test_dense_split_planes_16_byte_accesses(ctx, shape, dtype):
    """The reference's published N-D shapes in its split layout (float32 / float64 planes, pyfft/plan.py:26-35): one launch that moves
    16 bytes per lane and plane on either side.  The six-assertion protocol against numpy at a ragged batch
    (the last work-group holds fewer transforms than its tile) and at a batch of one; the interleaved plan of the same shape on the same
    numbers to rounding (same stage lists); the tiled kernel it replaces (MIFFT_DEBUG_ALT_ROWS = 7) within the same thresholds."""
    from pyfft_amd import _native as N
    hip = ctx.hip
    rd = numpy.dtype(dtype)
    double = rd == numpy.float64
    cdt = numpy.complex128 if double else numpy.complex64
    size = int(numpy.prod(shape))
    plan = ctx.getPlan(shape, dtype=dtype)
    assert len(plan.pass_list()) == 1 and plan.pass_list()[0].kind == N.PASS_ND, plan.pass_list()
    tile = 4096 if not double else 2048
    batch = 3 if size >= tile else (tile // size) * 2 + 1
    run_protocol(ctx, shape, dtype, batch, seed=6500 + size % 83, check_oracle=size * batch <= 16384)
    run_protocol(ctx, shape, dtype, 1, seed=6501, check_oracle=False)
    # against the interleaved plan on the same numbers
    rng = numpy.random.default_rng(6502)
    re = rng.standard_normal(size * batch).astype(rd)
    im = rng.standard_normal(size * batch).astype(rd)
    a_re, a_im = hip.to_gpu(re), hip.to_gpu(im)
    b_re, b_im = hip.DeviceArray((size * batch,), rd), hip.DeviceArray((size * batch,), rd)
    plan.execute(a_re, a_im, b_re, b_im, batch=batch)
    got = b_re.get().astype(numpy.complex128) + 1j * b_im.get()
    twin = ctx.getPlan(shape, dtype=cdt)
    c = hip.to_gpu((re + 1j * im).astype(cdt))
    twin.execute(c, batch=batch)
    want = c.get().astype(numpy.complex128)
    assert numpy.abs(want - got).sum() / numpy.abs(want).sum() < (1e-14 if double else 5e-7)
    N.check(N.lib.mifft_debug_set(N.DEBUG_ALT_ROWS, 7), "debug_set")
    try:
        old = ctx.getPlan(shape, dtype=dtype)
        if len(old.pass_list()) == 1:                      # (fp32 (16, 16, 128) and 32^3 are two launches without the dense kernel)
            old.execute(a_re, a_im, b_re, b_im, batch=batch)
            other = b_re.get().astype(numpy.complex128) + 1j * b_im.get()
            assert numpy.abs(other - got).sum() / numpy.abs(got).sum() < (1e-14 if double else 5e-7)
    finally:
        N.check(N.lib.mifft_debug_set(N.DEBUG_ALT_ROWS, 0), "debug_set")

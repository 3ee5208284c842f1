"""Round-4 GPU tests: the persistent two-pair kernel of the cache-sized cubes (128^3), the per-XCD work lists as a default
strategy, the sequential single-launch form of tiny batches, launches without memset / copy-back (two alternating counter sets,
pinned error word), the device properties the planner sizes everything from, Plan(stream=, context=i)."""
import ctypes
import os

import numpy
import pytest

import pyfft_oracle as oracle
from test_errors_gpu import run_protocol

pytestmark = pytest.mark.gpu
# the driver's GPU step has a time limit: the cases that repeat a kernel family on one more shape run with the soak switch PYFFT_AMD_SWEEP
_SOAK = bool(os.environ.get("PYFFT_AMD_SWEEP"))


def _noise(rng, count, dtype):
    """`count` N(0, 1) numbers: a seeded block of 2^22 + 17 draws, repeated (the period is no multiple of any transform size, so every
    transform of a batch sees different numbers; drawing 300 MiB afresh for every case took most of the suite's time)."""
    blk = rng.standard_normal(min(int(count), (1 << 22) + 17)).astype(dtype)
    return numpy.resize(blk, int(count))


def _test_data(shape, dtype, batch, seed):
    """Interleaved test data of `batch` transforms (the layout of oracle.get_test_data: the first axis times batch) from tiled noise."""
    rng = numpy.random.default_rng(seed)
    dtype = numpy.dtype(dtype)
    fdt = numpy.float32 if dtype == numpy.complex64 else numpy.float64
    full = [int(v) for v in (shape if isinstance(shape, tuple) else (shape,))]
    full[0] *= batch
    count = int(numpy.prod(full))
    out = numpy.empty(count, dtype)
    out.real = _noise(rng, count, fdt)
    out.imag = _noise(rng, count, fdt)
    return out.reshape(full)


def _execute(ctx, shape, dtype, batch, data, inplace=False, inverse=False, expect=None):
    plan = ctx.getPlan(shape, dtype=dtype)
    if expect is not None:
        assert plan.strategy(batch)[0] == expect, plan.strategy(batch)
    a = ctx.toGpu(data)
    if inplace:
        plan.execute(a, batch=batch, inverse=inverse)
        return a.get()
    b = ctx.allocate(data.shape, data.dtype)
    plan.execute(a, b, batch=batch, inverse=inverse)
    assert numpy.array_equal(a.get(), data), "an out-of-place execute touched its input"
    return b.get()


def test_device_properties_describe_the_memory_system(ctx):
    """mifft_device_props carries what the planner needs (include/mifft.h): on an MI355X 256 CUs in 8 XCDs with 4 MiB of L2
    each and the 256 MiB Infinity Cache -- read from the HSA agent, not hard-wired."""
    props = ctx.hip.device_props()
    m = ctx.hip.Machine.from_props(props)
    assert props.compute_units >= 1 and props.num_xcc >= 1 and props.llc_bytes >= 0
    if props.gcn_arch.decode().startswith("gfx950") and props.compute_units == 256:
        assert props.num_xcc == 8 and props.llc_bytes == 256 << 20 and props.l2_bytes == 4 << 20, (props.num_xcc, props.llc_bytes, props.l2_bytes)
        assert m.xcd_cooperative and m.ring_bytes == 224 << 20


# ---- persistent two-pair kernel: 128^3 -----------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype,batch", [(numpy.complex64, 19), (numpy.complex128, 10)], ids=lambda v: str(numpy.dtype(v).name) if isinstance(v, type) else str(v))
def test_fused_pair_cube_128(ctx, monkeypatch, dtype, batch):
    """(128, 128, 128) beyond the chain threshold: both pass pairs of every transform in ONE persistent launch
    (mifft_launch_fused_pair).  Same tile arithmetic as the two plain pair launches -> the bits of the chain; in place ==
    out of place; the reference's accuracy thresholds against numpy (test/test_errors.py:20-23) on sampled transforms; inverse
    round trip.  Published shape: doc/source/index.rst:373."""
    shape = (128, 128, 128)
    n = 128 ** 3
    cdt = numpy.dtype(dtype)
    tol, tol_max = (1.1e-6, 1e-5) if cdt == numpy.complex64 else (1e-11, 1e-10)
    rng = numpy.random.default_rng(77)
    data = (rng.standard_normal((batch * 128, 128, 128)) + 1j * rng.standard_normal((batch * 128, 128, 128))).astype(cdt)
    monkeypatch.setenv("PYFFT_AMD_STRATEGY", "chain")
    want = _execute(ctx, shape, dtype, batch, data, expect="chain")
    monkeypatch.setenv("PYFFT_AMD_STRATEGY", "auto")
    full = ctx.getPlan(shape, dtype=dtype).strategy(64)
    assert full[0] == "fusedp" and full[2] * n * cdt.itemsize <= 224 << 20, full      # the plan's own choice: a ring that fills the cache
    monkeypatch.setenv("PYFFT_AMD_FUSED_RING", "4,8" if cdt == numpy.complex64 else "2,4")   # (a ring the test's batch can fill twice)
    got = _execute(ctx, shape, dtype, batch, data, expect="fusedp")
    assert numpy.array_equal(want, got), "persistent two-pair launch differs from the two plain pair launches"
    got_ip = _execute(ctx, shape, dtype, batch, data, inplace=True, expect="fusedp")
    assert numpy.array_equal(got, got_ip), "in place differs from out of place"
    for item in (0, batch // 2, batch - 1):
        sl = slice(item * 128, (item + 1) * 128)
        ref = numpy.fft.fftn(data[sl].astype(numpy.complex128))
        assert numpy.abs(ref - got[sl]).sum() / numpy.abs(ref).sum() < tol
        assert numpy.abs(ref - got[sl]).max() <= tol_max * numpy.abs(ref).max()
    back = _execute(ctx, shape, dtype, batch, got, inverse=True, expect="fusedp")
    assert oracle.difference(data, back, batch) < tol
    # many executes in a row alternate between the two counter sets; a forced ring / lag keeps the results
    monkeypatch.setenv("PYFFT_AMD_FUSED_RING", "3,5")
    plan = ctx.getPlan(shape, dtype=dtype)
    assert plan.strategy(batch)[:3] == ("fusedp", 3, 5)
    a, b = ctx.toGpu(data), ctx.allocate(data.shape, data.dtype)
    for _ in range(5):
        plan.execute(a, b, batch=batch, wait_for_finish=False)
    plan.finish()
    assert numpy.array_equal(b.get(), got)


# ---- per-XCD work lists as a default ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,batch", [(1 << 16, 1040), (1 << 17, 520)], ids=str)
def test_per_xcd_lists(ctx, monkeypatch, n, batch):
    """The fused kernel with one work list per XCD (mifft_launch_fused2x; pyfft/kernel.py:259-283 chain semantics), on request
    (within round 4 the 32-column tiles of the global list overtook it at 2^17): the bits of the chain, in place == out of place; a
    batch that is not a multiple of 8 leaves the lists uneven, work stealing drains them."""
    from pyfft_amd import _native as N
    if N.lib.mifft_has_feature(N.FEATURE_FUSED2X) != 1:
        pytest.skip("development strategy: not in the default build of libmifft.so (make DEV=1)")
    if not ctx.hip.Machine.from_props(ctx.hip.device_props()).xcd_cooperative:
        pytest.skip("needs 8 XCDs x 32 CUs")
    data = _test_data((n,), numpy.complex64, batch, 91)
    monkeypatch.setenv("PYFFT_AMD_STRATEGY", "chain")
    want = _execute(ctx, (n,), numpy.complex64, batch, data, expect="chain")
    monkeypatch.setenv("PYFFT_AMD_STRATEGY", "fusedx")
    monkeypatch.setenv("PYFFT_AMD_FUSEDX", "8,16")
    got = _execute(ctx, (n,), numpy.complex64, batch, data, expect="fused2x")
    assert numpy.array_equal(want, got)
    assert numpy.array_equal(_execute(ctx, (n,), numpy.complex64, batch, data, inplace=True, expect="fused2x"), got)
    odd = batch - 3
    got_odd = _execute(ctx, (n,), numpy.complex64, odd, data[:odd * n], expect="fused2x")
    assert numpy.array_equal(got_odd, want[:odd * n])
    monkeypatch.setenv("PYFFT_AMD_STRATEGY", "auto")
    assert ctx.getPlan((n,), dtype=numpy.complex64).strategy(batch)[0] == "fused2"           # the plan's own choice: 32-column tiles


def test_fused_ring_rule_2_19(ctx, monkeypatch):
    """2^19 = 1024 x 512 has 32 first-pass tiles per transform: the ring rule counts the lag in tiles (28 transforms, ring 56 =
    224 MiB), not in transforms (round 3: 14 / 28, four points lower) -- and the bits stay the chain's."""
    n, batch = 1 << 19, 130
    plan = ctx.getPlan((n,), dtype=numpy.complex64)
    if ctx.hip.Machine.from_props(ctx.hip.device_props()).llc_bytes == 256 << 20 and plan._context.compute_units == 256:
        assert plan.strategy(batch) == ("fused2", 28, 56, 512)
    data = _test_data((n,), numpy.complex64, batch, 92)
    monkeypatch.setenv("PYFFT_AMD_STRATEGY", "chain")
    want = _execute(ctx, (n,), numpy.complex64, batch, data, expect="chain")
    monkeypatch.setenv("PYFFT_AMD_STRATEGY", "auto")
    assert numpy.array_equal(_execute(ctx, (n,), numpy.complex64, batch, data, expect="fused2"), want)


# ---- tiny batches: the sequential work list (one launch instead of two) ------------------------------------------------------
@pytest.mark.parametrize("shape,dtype,batch", [((1024, 1024), numpy.complex64, 4), ((1 << 20,), numpy.complex64, 3), ((1 << 18,), numpy.complex64, 7),
                                               ((128, 128, 128), numpy.complex64, 2), ((128, 128, 128), numpy.complex128, 1),
                                               ((1024, 1024), numpy.complex128, 2), ((1 << 22,), numpy.complex64, 1)],
                         ids=lambda v: str(numpy.dtype(v).name) if isinstance(v, type) else str(v).replace(" ", ""))
def test_sequential_single_launch_of_tiny_batches(ctx, monkeypatch, shape, dtype, batch):
    """The reference's own benchmark protocol runs 32 MiB buffers (test/test_performance.py:11,22-30): there the two passes of a
    transform are two dependent launches.  The sequential work list runs them in ONE persistent launch (lag 0: every first-pass
    tile, then every second-pass tile); it must give the chain's bits, in place and out of place, forward and inverse.  Measured
    slower than the two launches: part of `make DEV=1` builds of the library only."""
    from pyfft_amd import _native as N
    if N.lib.mifft_has_feature(N.FEATURE_SEQUENTIAL_LIST) != 1:
        pytest.skip("development form: not in the default build of libmifft.so (make DEV=1)")
    data = _test_data(shape, dtype, batch, 93)
    monkeypatch.setenv("PYFFT_AMD_SMALL_FUSED", "0")
    want = _execute(ctx, shape, dtype, batch, data, expect="chain")
    monkeypatch.setenv("PYFFT_AMD_SMALL_FUSED", "1")
    plan = ctx.getPlan(shape, dtype=dtype)
    st = plan.strategy(batch)
    assert st[0] in ("fused2", "fusedp") and st[1] == 0 and st[2] == batch, st
    got = _execute(ctx, shape, dtype, batch, data)
    if len(shape) == 2:
        # the 2-D persistent form is two TRANSPOSING passes, the chain a ROW pass and a strided one: the same transform in another
        # operation order (as in test_fused_2d_1024)
        assert oracle.difference(want, got, batch) < (5e-7 if numpy.dtype(dtype) == numpy.complex64 else 1e-14)
    else:
        assert numpy.array_equal(want, got)
    assert numpy.array_equal(_execute(ctx, shape, dtype, batch, data, inplace=True), got)
    tol = 1.1e-6 if numpy.dtype(dtype) == numpy.complex64 else 1e-11
    back = _execute(ctx, shape, dtype, batch, got, inverse=True)
    assert oracle.difference(data, back, batch) < tol
    ref = oracle.numpy_fft(numpy.fft.fftn, data, batch)
    assert oracle.difference(ref, got, batch) < tol


# ---- launches without memset / copy-back -----------------------------------------------------------------------------------
def test_alternating_counter_sets_and_memset_form_agree(ctx, monkeypatch):
    """A plan's persistent launches alternate between two counter sets (each launch zeroes the other one: mifft_fused_sync), so
    no memset node precedes a launch; the round-3 form (one set, zeroed by the call) is kept behind a switch.  Both give the same
    bits over many back-to-back executes, with a batch change in between (new sets) and alternating directions."""
    n = 1 << 20
    data = _test_data((n,), numpy.complex64, 70, 94)
    outs = {}
    for memset in ("", "1"):
        if memset:
            monkeypatch.setenv("PYFFT_AMD_FUSED_MEMSET", memset)
        plan = ctx.getPlan((n,), dtype=numpy.complex64, wait_for_finish=False)
        res = []
        for batch in (70, 60, 70):
            assert plan.strategy(batch)[0] == "fused2"
            a, b = ctx.toGpu(data[:batch * n]), ctx.allocate((batch * n,), numpy.complex64)
            for rep in range(5):
                plan.execute(a, b, batch=batch)
                plan.execute(b, a, batch=batch, inverse=True)         # back to the data (to rounding)
            plan.execute(a, b, batch=batch)
            plan.finish()
            res.append(b.get())
        outs[memset] = res
    for x, y in zip(outs[""], outs["1"]):
        assert numpy.array_equal(x, y)
    ref = oracle.numpy_fft(numpy.fft.fft, data[:2 * n], 2)
    assert oracle.difference(ref, outs[""][0][:2 * n], 2) < 2e-6      # (eleven transforms deep)


def test_plan_with_stream_and_context_index(ctx):
    """Plan(stream=s, context=i): the device comes from `context` also when a stream is given (cuda.py:121-134); the plan is
    asynchronous by default and guarded for device i.  With several GPUs the last one is used from device 0."""
    hip = ctx.hip
    N = hip.N
    ndev = hip.device_count()
    dev = ndev - 1
    cur = ctypes.c_int()
    N.check(N.lib.mifft_get_device(ctypes.byref(cur)), "get")
    home = cur.value
    N.check(N.lib.mifft_set_device(dev), "set")
    stream = hip.Stream()
    data = _test_data((8192,), numpy.complex64, 3, 95)
    a = ctx.toGpu(data)
    N.check(N.lib.mifft_set_device(home), "set")
    plan = ctx.getPlan((8192,), dtype=numpy.complex64, stream=stream, context=dev)
    assert plan._context.device == dev and plan._context._guard and plan._wait_for_finish is False
    assert plan.execute(a, batch=3) is stream
    plan.finish()
    N.check(N.lib.mifft_get_device(ctypes.byref(cur)), "get")
    assert cur.value == home
    N.check(N.lib.mifft_set_device(dev), "set")
    got = a.get()
    N.check(N.lib.mifft_set_device(home), "set")
    assert oracle.difference(oracle.numpy_fft(numpy.fft.fft, data, 3), got, 3) < 1.1e-6

    class FakeTorchStream(object):                 # a stream that knows its device (torch.cuda.Stream.device_index)
        cuda_stream = stream.handle
        device_index = dev + 1
    with pytest.raises(ValueError, match="stream belongs to device"):
        ctx.getPlan((8192,), dtype=numpy.complex64, stream=FakeTorchStream(), context=dev)


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


# ---- rectangular 2-D shapes on the fused kernel ---------------------------------------------------------------------------------
RECT_2D_CASES = [((512, 1024), 66), ((1024, 512), 113), ((1024, 2048), 18), ((2048, 1024), 29), ((512, 2048), 33), ((2048, 512), 57)]


@pytest.mark.parametrize("shape,batch", RECT_2D_CASES, ids=str)   # ((512, 2048) runs the kernel on request only: pipelined is faster)
def test_fused_2d_rectangles(ctx, monkeypatch, shape, batch):
    """(ny, nx) in {512, 1024, 2048}^2 with ny != nx, fp32 interleaved, beyond the chain threshold: one persistent launch of two
    transposing passes (round 3: squares only; pyfft/kernel.mako:857-874 vertical mode, plan.py:135-171).  The reference's
    thresholds against numpy on sampled transforms, input untouched, in place == out of place, inverse round trip, and the
    chain's result within fp32 rounding (another operation order: ROW + strided COL)."""
    ny, nx = shape
    data = _test_data(shape, numpy.complex64, batch, 1100 + ny // 512 + nx // 128)
    monkeypatch.setenv("PYFFT_AMD_STRATEGY", "auto" if shape != (512, 2048) else "fused")
    got = _execute(ctx, shape, numpy.complex64, batch, data, expect="fused2")
    assert numpy.array_equal(_execute(ctx, shape, numpy.complex64, batch, data, inplace=True, expect="fused2"), got)
    for item in (0, batch // 2, batch - 1):
        sl = slice(item * ny, (item + 1) * ny)
        ref = numpy.fft.fft2(data[sl].astype(numpy.complex128))
        assert numpy.abs(ref - got[sl]).sum() / numpy.abs(ref).sum() < 1.1e-6
        assert numpy.abs(ref - got[sl]).max() <= 1e-5 * numpy.abs(ref).max()
    back = _execute(ctx, shape, numpy.complex64, batch, got, inverse=True, expect="fused2")
    assert oracle.difference(data, back, batch) < 1.1e-6
    monkeypatch.setenv("PYFFT_AMD_STRATEGY", "chain")
    want = _execute(ctx, shape, numpy.complex64, batch, data, expect="chain")
    assert oracle.difference(want, got, batch) < 5e-7
    # split planes: the row-first kernel where both sides are <= 1024 (test_fused_2d_split_row_first), else the pipelined chunks
    monkeypatch.setenv("PYFFT_AMD_STRATEGY", "auto")
    assert ctx.getPlan(shape, dtype=numpy.float32).strategy(batch)[0] == ("fused2" if max(shape) <= 1024 else "pipelined")


# ---- f4 tails: smooth N-D shapes in one launch, Bluestein rows up to 5000 points in one launch ----------------------------------
@pytest.mark.parametrize("dtype", [numpy.complex64, numpy.complex128], ids=["c64", "c128"])
@pytest.mark.parametrize("shape,batch", [((100, 100), 5), ((60, 60), 7), ((30, 20, 10), 3), ((12, 20), 301), ((6, 10, 14), 33), ((100, 64), 2),
                                         ((70, 70), 3), ((9, 1, 25), 11), ((15, 16), 64)], ids=str)
def test_smooth_nd_single_launch(ctx, monkeypatch, shape, dtype, batch):
    """Every axis a smooth length and the transform inside one tile (csrc/fft_mixed_nd.hip; the reference's TODO.txt:8): ONE launch
    against numpy with the reference's thresholds, out of place (input untouched), the inverse in place, ragged last work-group,
    and against the round-3 form (one launch per axis) -- the same butterflies in the same order, so the same bits."""
    from test_round2_gpu import _run_generic
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
    from test_round2_gpu import _run_generic
    N = ctx.hip.N
    prec = N.F32 if numpy.dtype(dtype) == numpy.complex64 else N.F64
    m = ctypes.c_int32(0)
    one = N.lib.mifft_bluestein_padded(prec, n, ctypes.byref(m)) == 0
    assert one == (n <= (5000 if prec == N.F32 else 2500))
    plan = ctx.getPlan((n,), dtype=dtype, any_size=True)
    smooth = N.lib.mifft_mixed_supported(prec, n) == 0 or plan._direct_long is not None
    assert plan._direct_blue == (one and not smooth)
    _run_generic(ctx, (n,), dtype, batch, seed=n)


# ---- fp64 2^21 / 2^22 on the persistent kernel (stage-chain strided tiles) ------------------------------------------------------
@pytest.mark.parametrize("n,batch", [(1 << 22, 6), (1 << 21, 14)], ids=str)
def test_fused_long_fp64(ctx, monkeypatch, n, batch):
    """fp64 N = 2^22 = 2048 x 2048 and 2^21 = 2048 x 1024 beyond the chain threshold: both passes in one persistent launch on the
    8-column stage-chain tiles (csrc/fft_fusedx_f64.hip; round 3: two launches per 64 MiB chunk).  2^22 runs the chain's own two
    tile kinds -> the chain's bits; 2^21 runs its 1024-point pass on 16-column stage-chain tiles instead of the 512-thread ones ->
    the same transform in another operation order.  numpy on sampled transforms (reference thresholds), in place, inverse."""
    rng = numpy.random.default_rng(n % 1000)
    blk = (rng.standard_normal((2, n)) + 1j * rng.standard_normal((2, n))).astype(numpy.complex128)
    data = numpy.concatenate([blk[i % 2] for i in range(batch)])
    monkeypatch.setenv("PYFFT_AMD_STRATEGY", "chain")
    want = _execute(ctx, (n,), numpy.complex128, batch, data, expect="chain")
    auto = "auto" if n == 1 << 21 else "fused"        # (2^22: the kernel exists and is tested, the plan prefers the pipelined chunks)
    monkeypatch.setenv("PYFFT_AMD_STRATEGY", auto)
    got = _execute(ctx, (n,), numpy.complex128, batch, data, expect="fused2")
    if n == 1 << 22:
        assert numpy.array_equal(want, got)
    else:
        assert oracle.difference(want, got, batch) < 1e-14
    assert numpy.array_equal(_execute(ctx, (n,), numpy.complex128, batch, data, inplace=True, expect="fused2"), got)
    if n == 1 << 22:
        monkeypatch.setenv("PYFFT_AMD_STRATEGY", "auto")
        assert ctx.getPlan((n,), dtype=numpy.complex128).strategy(64)[0] == "pipelined"
        monkeypatch.setenv("PYFFT_AMD_STRATEGY", auto)
    for item in (0, 1, batch - 1):
        ref = numpy.fft.fft(blk[item % 2])
        g = got[item * n:(item + 1) * n]
        assert numpy.abs(ref - g).sum() / numpy.abs(ref).sum() < 1e-11
        assert numpy.abs(ref - g).max() <= 1e-10 * numpy.abs(ref).max()
    back = _execute(ctx, (n,), numpy.complex128, batch, got, inverse=True, expect="fused2")
    assert oracle.difference(data, back, batch) < 1e-11
    # split planes have no such kernel: the pipelined chunks
    monkeypatch.setenv("PYFFT_AMD_STRATEGY", "auto")
    assert ctx.getPlan((n,), dtype=numpy.float64).strategy(batch)[0] in ("pipelined", "chain")


@pytest.mark.parametrize("dtype", [numpy.complex64, numpy.complex128], ids=["c64", "c128"])
@pytest.mark.parametrize("shape,batch", [((60, 60, 60), 2), ((24, 100, 100), 3), ((7, 90, 50), 5)], ids=str)
def test_smooth_3d_as_planes_and_lines(ctx, shape, dtype, batch):
    """3-D smooth shapes beyond one tile whose (y, x) planes fit one: the planes in ONE launch (they are more transforms of the
    2-D kernel), then the z lines -- two HBM round trips instead of three (round 3: one launch per axis).  numpy, reference
    thresholds, out of place with the input untouched, the normalised inverse in place."""
    from test_round2_gpu import _run_generic
    N = ctx.hip.N
    prec = N.F32 if numpy.dtype(dtype) == numpy.complex64 else N.F64
    z, y, x = shape
    plan = ctx.getPlan(shape, dtype=dtype, any_size=True)
    planes = N.lib.mifft_mixed_nd_supported(prec, x, y, z) != 0 and N.lib.mifft_mixed_nd_supported(prec, x, y, 1) == 0
    assert plan._direct_nd_planes == planes and not plan._direct_nd1 and plan._inner_plans() == []
    _run_generic(ctx, shape, dtype, batch, seed=sum(shape))


@pytest.mark.parametrize("n,batch,forced", [(1 << 19, 34, "auto"), (1 << 18, 66, "auto"), (1 << 17, 130, "auto"), (1 << 16, 260, "auto")], ids=str)
def test_fused_mid_sizes_fp64(ctx, monkeypatch, n, batch, forced):
    """fp64 N = 2^16 ... 2^18 (L0 >= L1 in {256, 512}) on the persistent two-pass kernel with the 256-thread two-phase tiles
    (`fft_fused2_kernel<double>`; round 3: pipelined chunks): the chain's tile code, so the chain's bits; 2^19 = 1024 x 512 on the
    512-thread tiles (`fft_fused3_kernel<double, 2, 1>`); in place; numpy on sampled
    transforms with the reference's fp64 thresholds (test/test_errors.py:20-23); inverse round trip."""
    data = _test_data((n,), numpy.complex128, batch, 97)
    monkeypatch.setenv("PYFFT_AMD_STRATEGY", "chain")
    want = _execute(ctx, (n,), numpy.complex128, batch, data, expect="chain")
    monkeypatch.setenv("PYFFT_AMD_STRATEGY", forced)
    monkeypatch.setenv("PYFFT_AMD_FUSED_RING", "8,16")          # (a ring this batch can fill twice)
    got = _execute(ctx, (n,), numpy.complex128, batch, data, expect="fused2")
    if n == 1 << 19:
        # 1024 x 512: the persistent kernel runs the 512-point pass on the 512-thread tiles (2 x 256 by decimation in time), the chain
        # on the 256-thread ones: the same transform in another operation order
        assert oracle.difference(want, got, batch) < 1e-14
    else:
        assert numpy.array_equal(want, got)
    assert numpy.array_equal(_execute(ctx, (n,), numpy.complex128, batch, data, inplace=True, expect="fused2"), got)
    for item in (0, batch // 2, batch - 1):
        ref = numpy.fft.fft(data[item * n:(item + 1) * n])
        g = got[item * n:(item + 1) * n]
        assert numpy.abs(ref - g).sum() / numpy.abs(ref).sum() < 1e-11
        assert numpy.abs(ref - g).max() <= 1e-10 * numpy.abs(ref).max()
    back = _execute(ctx, (n,), numpy.complex128, batch, got, inverse=True, expect="fused2")
    assert oracle.difference(data, back, batch) < 1e-11


@pytest.mark.parametrize("shape,batch", [((512, 512), 66), ((512, 1024), 34), ((1024, 512), 34)], ids=str)
def test_fused_2d_fp64_512_sides(ctx, monkeypatch, shape, batch):
    """fp64 2-D shapes with a 512-point side on the persistent 2-D kernels (round 4; round 3: 1024 x 1024 only): (512, 512) on the
    256-thread two-phase tiles, the rectangles on the 512-thread ones.  numpy with the reference's fp64 thresholds on sampled
    transforms, in place == out of place, inverse, and the chain's result to rounding (two transposing passes against ROW + COL)."""
    ny, nx = shape
    data = _test_data(shape, numpy.complex128, batch, 1200 + ny // 512 + nx // 256)
    monkeypatch.setenv("PYFFT_AMD_STRATEGY", "auto")
    monkeypatch.setenv("PYFFT_AMD_FUSED_RING", "8,16")
    got = _execute(ctx, shape, numpy.complex128, batch, data, expect="fused2")
    assert numpy.array_equal(_execute(ctx, shape, numpy.complex128, batch, data, inplace=True, expect="fused2"), got)
    for item in (0, batch // 2, batch - 1):
        sl = slice(item * ny, (item + 1) * ny)
        ref = numpy.fft.fft2(data[sl])
        assert numpy.abs(ref - got[sl]).sum() / numpy.abs(ref).sum() < 1e-11
        assert numpy.abs(ref - got[sl]).max() <= 1e-10 * numpy.abs(ref).max()
    back = _execute(ctx, shape, numpy.complex128, batch, got, inverse=True, expect="fused2")
    assert oracle.difference(data, back, batch) < 1e-11
    monkeypatch.setenv("PYFFT_AMD_STRATEGY", "chain")
    want = _execute(ctx, shape, numpy.complex128, batch, data, expect="chain")
    assert oracle.difference(want, got, batch) < 1e-14


WIDE_TILE_CASES = [(1 << 16, 1040), (1 << 17, 530), (1 << 18, 161)] + ([(1 << 18, 260)] if _SOAK else [])      # (soak: a second batch of one size)


@pytest.mark.parametrize("n,batch", WIDE_TILE_CASES, ids=str)
def test_wide_tiles_fp32_mid_sizes(ctx, monkeypatch, n, batch):
    """fp32 N = 2^16 ... 2^18 in the persistent kernel on 32-column tiles (csrc/fft_col2w.hpp: a thread owns two adjacent columns,
    16-byte lanes, 256-byte row segments) by the plan's own choice: the bits of the chain (same butterflies, same table factors) and
    of the 16-column tiles, in place, numpy with the reference's thresholds, inverse round trip; a batch that fills the ring only
    once takes half the pipeline instead of falling back to the chunks."""
    N = ctx.hip.N
    data = _test_data((n,), numpy.complex64, batch, 98)
    monkeypatch.setenv("PYFFT_AMD_STRATEGY", "chain")
    want = _execute(ctx, (n,), numpy.complex64, batch, data, expect="chain")
    monkeypatch.setenv("PYFFT_AMD_STRATEGY", "auto")
    got = _execute(ctx, (n,), numpy.complex64, batch, data, expect="fused2")
    assert numpy.array_equal(want, got)
    assert numpy.array_equal(_execute(ctx, (n,), numpy.complex64, batch, data, inplace=True, expect="fused2"), got)
    if N.lib.mifft_has_feature(N.FEATURE_AB_FORMS) == 1:        # (the 16-column persistent form of these lengths: `make DEV=1` builds)
        N.check(N.lib.mifft_debug_set(N.DEBUG_NARROW_TILES, 1), "debug_set")
        try:
            monkeypatch.setenv("PYFFT_AMD_STRATEGY", "fused")
            narrow = _execute(ctx, (n,), numpy.complex64, batch, data, expect="fused2")
        finally:
            N.check(N.lib.mifft_debug_set(N.DEBUG_NARROW_TILES, 0), "debug_set")
        assert numpy.array_equal(narrow, got)
    monkeypatch.setenv("PYFFT_AMD_STRATEGY", "auto")
    for item in (0, batch // 2, batch - 1):
        ref = numpy.fft.fft(data[item * n:(item + 1) * n].astype(numpy.complex128))
        g = got[item * n:(item + 1) * n]
        assert numpy.abs(ref - g).sum() / numpy.abs(ref).sum() < 1.1e-6
        assert numpy.abs(ref - g).max() <= 1e-5 * numpy.abs(ref).max()
    back = _execute(ctx, (n,), numpy.complex64, batch, got, inverse=True, expect="fused2")
    assert oracle.difference(data, back, batch) < 1.1e-6


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
    from test_round2_gpu import _numpy_tiles
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


# ---- 2-D shapes with a 256-point axis on the persistent kernels (second batch of round 4) ---------------------------------------
@pytest.mark.parametrize("dtype", [numpy.complex64, numpy.complex128], ids=["c64", "c128"])
@pytest.mark.parametrize("shape,batch", [((256, 256), 530), ((256, 512), 270), ((512, 256), 265), ((256, 1024), 140), ((1024, 256), 133)], ids=str)
def test_fused_2d_256_sides(ctx, monkeypatch, shape, batch, dtype):
    """(ny, nx) with a 256-point axis beyond the chain threshold, interleaved: fp32 next to a side <= 1024 on the 32-column tiles
    (fft_fused2dw_kernel), fp64 next to a side <= 512 on the 256-thread two-phase tiles -- one persistent launch of two transposing
    passes instead of ROW + strided COL per cache-sized chunk (pyfft/plan.py:135-171).  numpy with the reference's thresholds on
    sampled transforms, input untouched, in place == out of place, inverse, and the chain's result to rounding."""
    ny, nx = shape
    cd = numpy.dtype(dtype)
    f64 = cd == numpy.complex128
    if f64 and max(shape) > 512:
        assert ctx.getPlan(shape, dtype=dtype).strategy(batch)[0] in ("pipelined", "chain")
        return
    if f64:
        batch = batch // 2 + 1
    eps, mx = (1e-11, 1e-10) if f64 else (1.1e-6, 1e-5)
    data = _test_data(shape, cd, batch, 1300 + ny // 256 + nx // 64)
    monkeypatch.setenv("PYFFT_AMD_STRATEGY", "auto")
    got = _execute(ctx, shape, cd, batch, data, expect="fused2")
    inp = _execute(ctx, shape, cd, batch, data, inplace=True, expect="fused2")
    if ctx.getPlan(shape, dtype=cd)._oop_nd is not None:
        # round 5: (256, 256) fp32 runs ONE launch with four work-groups per transform when it is out of place (csrc/fft_nd2z.hpp) and the
        # persistent kernel in place: two factorisations of one transform, equal to rounding (the reference's protocol asks no more:
        # test/test_errors.py:97-103)
        assert oracle.difference(inp, got, batch) < 5e-7
        got = inp             # (what follows is about the persistent kernel)
    else:
        assert numpy.array_equal(inp, got)
    for item in (0, 1, batch // 2, batch - 1):
        sl = slice(item * ny, (item + 1) * ny)
        ref = numpy.fft.fft2(data[sl].astype(numpy.complex128))
        assert numpy.abs(ref - got[sl]).sum() / numpy.abs(ref).sum() < eps
        assert numpy.abs(ref - got[sl]).max() <= mx * numpy.abs(ref).max()
    back = _execute(ctx, shape, cd, batch, got, inverse=True, inplace=True, expect="fused2")
    assert oracle.difference(data, back, batch) < eps
    monkeypatch.setenv("PYFFT_AMD_STRATEGY", "chain")
    want = _execute(ctx, shape, cd, batch, data, expect="chain")
    assert oracle.difference(want, got, batch) < (1e-14 if f64 else 5e-7)


# ---- split-complex fp32 on the persistent 1-D kernel: sibling tiles per item ------------------------------------------------------
def _execute_split(ctx, shape, rdtype, batch, re, im, inplace=False, inverse=False, expect=None):
    plan = ctx.getPlan(shape, dtype=rdtype)
    if expect is not None:
        assert plan.strategy(batch)[0] == expect, plan.strategy(batch)
    a_re, a_im = ctx.toGpu(re), ctx.toGpu(im)
    if inplace:
        plan.execute(a_re, a_im, batch=batch, inverse=inverse)
        return a_re.get(), a_im.get()
    b_re, b_im = ctx.allocate(re.shape, re.dtype), ctx.allocate(im.shape, im.dtype)
    plan.execute(a_re, a_im, b_re, b_im, batch=batch, inverse=inverse)
    assert numpy.array_equal(a_re.get(), re) and numpy.array_equal(a_im.get(), im), "an out-of-place execute touched its input"
    return b_re.get(), b_im.get()


SPLIT_1D_CASES = [(1 << 16, 1100, "fused2"), (1 << 17, 515, "fused2"), (1 << 18, 259, "fused2"), (1 << 19, 130, "fused2"), (1 << 20, 70, "fused2")]


@pytest.mark.parametrize("n,batch,expect", SPLIT_1D_CASES, ids=str)
def test_split_planes_on_per_xcd_lists(ctx, monkeypatch, n, batch, expect):
    """float32 planes (the reference's split layout, pyfft/plan.py:10-63 dtype rule) on the persistent 1-D kernels by the plan's own
    choice: the two 16-column tiles that share every 128-byte line of a plane run in one 512-thread work-group, interleaved at lane
    level (fft_fused2s_kernel on the global list); with PYFFT_AMD_SPLIT_FUSEDX the per-XCD lists, where siblings share an L2 (2^16
    ... 2^18).  The bits of the chain (same
    tiles, same order of operations), in place == out of place, numpy with the reference's thresholds on sampled transforms, the
    inverse round trip, batches that are no multiple of 8 (lists of unequal length)."""
    rng = numpy.random.default_rng(1400 + n % 97)
    re = _noise(rng, n * batch, numpy.float32)
    im = _noise(rng, n * batch, numpy.float32)
    monkeypatch.setenv("PYFFT_AMD_STRATEGY", "chain")
    want = _execute_split(ctx, (n,), numpy.float32, batch, re, im, expect="chain")
    monkeypatch.setenv("PYFFT_AMD_STRATEGY", "auto")
    got = _execute_split(ctx, (n,), numpy.float32, batch, re, im, expect=expect)
    assert numpy.array_equal(want[0], got[0]) and numpy.array_equal(want[1], got[1])
    inp = _execute_split(ctx, (n,), numpy.float32, batch, re, im, inplace=True, expect=expect)
    assert numpy.array_equal(inp[0], got[0]) and numpy.array_equal(inp[1], got[1])
    for item in (0, batch // 2, batch - 1):
        sl = slice(item * n, (item + 1) * n)
        ref = numpy.fft.fft(re[sl].astype(numpy.float64) + 1j * im[sl].astype(numpy.float64))
        g = got[0][sl] + 1j * got[1][sl]
        assert numpy.abs(ref - g).sum() / numpy.abs(ref).sum() < 1.1e-6
        assert numpy.abs(ref - g).max() <= 1e-5 * numpy.abs(ref).max()
    back = _execute_split(ctx, (n,), numpy.float32, batch, got[0], got[1], inverse=True, expect=expect)
    x = re + 1j * im
    assert numpy.abs((back[0] + 1j * back[1]) - x).sum() / numpy.abs(x).sum() < 1.1e-6
    from pyfft_amd import _native as N
    if n <= (1 << 18) and N.lib.mifft_has_feature(N.FEATURE_FUSED2X) == 1:      # (`make DEV=1` builds)
        monkeypatch.setenv("PYFFT_AMD_SPLIT_FUSEDX", "1")
        lists = _execute_split(ctx, (n,), numpy.float32, batch, re, im, expect="fused2x")
        assert numpy.array_equal(lists[0], got[0]) and numpy.array_equal(lists[1], got[1])


# ---- persistent two-pair kernel for 3-D shapes with 64- and 128-point axes (csrc/fft_fusedp2.hip) --------------------------------
PAIR_SMALL_AXES_CASES = [((64, 64, 64), 141), ((64, 128, 128), 59), ((128, 128, 64), 57), ((128, 64, 128), 61),
                         ((64, 128, 64), 115), ((64, 64, 128), 117), ((128, 64, 64), 113)]


@pytest.mark.parametrize("dtype", [numpy.complex64, numpy.complex128], ids=["c64", "c128"])
@pytest.mark.parametrize("shape,batch", PAIR_SMALL_AXES_CASES, ids=str)
def test_fused_pair_small_axes(ctx, monkeypatch, shape, batch, dtype):
    """3-D shapes whose chain is one plane pass + one strided z pass (pyfft/plan.py:160-167: one chain per axis) and that have a
    persistent two-pair kernel: beyond the chain threshold the plan factors the y axis R0 x R1 FOR THAT LAUNCH ALONE (four passes as
    two tile kinds, the buffer between them a ring in the last-level cache).  The reference's thresholds against numpy on sampled
    transforms, input untouched, in place == out of place, inverse round trip, the chain's result to rounding (another operation
    order), and the A/B switch that keeps such shapes on their chain."""
    cd = numpy.dtype(dtype)
    f64 = cd == numpy.complex128
    if f64:
        batch = batch // 2 + 1
    nz, ny, nx = shape
    n = nz * ny * nx
    eps, mx = (1e-11, 1e-10) if f64 else (1.1e-6, 1e-5)
    plan = ctx.getPlan(shape, dtype=cd)
    assert plan._pair_alt is not None and len(plan._kernels) == 2 and not plan._paired
    assert plan.strategy(2)[0] == "chain"
    data = _test_data(shape, cd, batch, 1500 + nz + nx // 64)
    monkeypatch.setenv("PYFFT_AMD_STRATEGY", "auto")
    got = _execute(ctx, shape, cd, batch, data, expect="fusedp")
    assert numpy.array_equal(_execute(ctx, shape, cd, batch, data, inplace=True, expect="fusedp"), got)
    for item in (0, 1, batch // 2, batch - 1):
        ref = numpy.fft.fftn(data.reshape((batch,) + shape)[item].astype(numpy.complex128)).reshape(-1)
        g = got.reshape(batch, n)[item]
        assert numpy.abs(ref - g).sum() / numpy.abs(ref).sum() < eps
        assert numpy.abs(ref - g).max() <= mx * numpy.abs(ref).max()
    back = _execute(ctx, shape, cd, batch, got, inverse=True, expect="fusedp")
    assert oracle.difference(data, back, batch) < eps
    monkeypatch.setenv("PYFFT_AMD_STRATEGY", "chain")
    want = _execute(ctx, shape, cd, batch, data, expect="chain")
    assert oracle.difference(want, got, batch) < (1e-14 if f64 else 5e-7)
    monkeypatch.setenv("PYFFT_AMD_STRATEGY", "auto")
    monkeypatch.setenv("PYFFT_AMD_NO_FUSEDP_ALT", "1")
    assert ctx.getPlan(shape, dtype=cd).strategy(batch)[0] == "pipelined"


# (every shape of {64, 128}^3 has its own tile pair in both precisions: every instance a default plan can select runs in the default suite)
_PAIR_SPLIT_CASES = [((64, 64, 64), numpy.float32, 141), ((64, 128, 128), numpy.float32, 59), ((128, 64, 64), numpy.float32, 113),
                     ((128, 128, 128), numpy.float64, 15), ((128, 128, 64), numpy.float64, 30), ((128, 64, 128), numpy.float64, 31),
                     ((64, 128, 64), numpy.float32, 115), ((64, 64, 128), numpy.float32, 117), ((64, 64, 64), numpy.float64, 71),
                     ((128, 64, 64), numpy.float64, 57), ((64, 128, 128), numpy.float64, 29), ((128, 128, 128), numpy.float32, 29),
                     ((128, 128, 64), numpy.float32, 57), ((128, 64, 128), numpy.float32, 61), ((64, 128, 64), numpy.float64, 58), ((64, 64, 128), numpy.float64, 59)]


@pytest.mark.parametrize("shape,rdtype,batch", _PAIR_SPLIT_CASES, ids=str)
def test_fused_pair_split_planes(ctx, monkeypatch, shape, rdtype, batch):
    """Split-complex user buffers (the reference's float32 / float64 dtypes, pyfft/plan.py:10-63) on the persistent two-pair kernel:
    the XY tiles read the re / im planes, the YZ tiles write them, the ring between them is interleaved.  Every shape of
    {64, 128}^3 in both precisions.  numpy with the reference's thresholds
    on sampled transforms, planes untouched, in place == out of place, inverse round trip, the chain's result to rounding."""
    rd = numpy.dtype(rdtype)
    f64 = rd == numpy.float64
    eps, mx = (1e-11, 1e-10) if f64 else (1.1e-6, 1e-5)
    n = shape[0] * shape[1] * shape[2]
    rng = numpy.random.default_rng(1600 + sum(shape))
    re = _noise(rng, n * batch, rd)
    im = _noise(rng, n * batch, rd)
    monkeypatch.setenv("PYFFT_AMD_STRATEGY", "auto")
    assert ctx.getPlan(shape, dtype=rd)._pair_alt is not None
    got = _execute_split(ctx, shape, rd, batch, re, im, expect="fusedp")
    inp = _execute_split(ctx, shape, rd, batch, re, im, inplace=True, expect="fusedp")
    assert numpy.array_equal(inp[0], got[0]) and numpy.array_equal(inp[1], got[1])
    for item in (0, 1, batch // 2, batch - 1):
        sl = slice(item * n, (item + 1) * n)
        ref = numpy.fft.fftn((re[sl].astype(numpy.float64) + 1j * im[sl].astype(numpy.float64)).reshape(shape)).reshape(-1)
        g = got[0][sl] + 1j * got[1][sl]
        assert numpy.abs(ref - g).sum() / numpy.abs(ref).sum() < eps
        assert numpy.abs(ref - g).max() <= mx * numpy.abs(ref).max()
    back = _execute_split(ctx, shape, rd, batch, got[0], got[1], inverse=True, expect="fusedp")
    x = re + 1j * im
    assert numpy.abs((back[0] + 1j * back[1]) - x).sum() / numpy.abs(x).sum() < eps
    monkeypatch.setenv("PYFFT_AMD_STRATEGY", "chain")
    want = _execute_split(ctx, shape, rd, batch, re, im, expect="chain")
    d = numpy.abs((want[0] - got[0]) + 1j * (want[1] - got[1])).sum() / numpy.abs(want[0] + 1j * want[1]).sum()
    assert d < (1e-14 if f64 else 5e-7)


# ---- split-complex fp32 2-D plans on the row-first persistent kernel (csrc/fft_fused2r.hpp) ------------------------------------------
SPLIT_2D_ROWFIRST_CASES = [((1024, 1024), 37), ((512, 512), 140), ((256, 256), 530), ((512, 1024), 70), ((1024, 256), 131), ((256, 512), 261),
                           ((1024, 512), 67), ((256, 1024), 135), ((512, 256), 259)]


@pytest.mark.parametrize("shape,batch", SPLIT_2D_ROWFIRST_CASES, ids=str)   # ((256, 256): on request only)
def test_fused_2d_split_row_first(ctx, monkeypatch, shape, batch):
    """float32 planes, 2-D, beyond the chain threshold: ROW x from the planes and COL y to the planes on the persistent work list (the
    chain's own order, pyfft/plan.py:135-171, instead of two transposing passes whose 16-column tiles read half lines of the planes).
    numpy with the reference's thresholds on sampled transforms, planes untouched, in place == out of place, inverse round trip, the
    chain's result to rounding, and the A/B switch back to the pipelined chunks."""
    ny, nx = shape
    n = ny * nx
    rng = numpy.random.default_rng(1700 + ny // 256 + nx // 64)
    re = _noise(rng, n * batch, numpy.float32)
    im = _noise(rng, n * batch, numpy.float32)
    monkeypatch.setenv("PYFFT_AMD_STRATEGY", "auto" if shape != (256, 256) else "fused")
    got = _execute_split(ctx, shape, numpy.float32, batch, re, im, expect="fused2")
    inp = _execute_split(ctx, shape, numpy.float32, batch, re, im, inplace=True, expect="fused2")
    assert numpy.array_equal(inp[0], got[0]) and numpy.array_equal(inp[1], got[1])
    for item in (0, 1, batch // 2, batch - 1):
        sl = slice(item * n, (item + 1) * n)
        ref = numpy.fft.fft2((re[sl].astype(numpy.float64) + 1j * im[sl].astype(numpy.float64)).reshape(shape)).reshape(-1)
        g = got[0][sl] + 1j * got[1][sl]
        assert numpy.abs(ref - g).sum() / numpy.abs(ref).sum() < 1.1e-6
        assert numpy.abs(ref - g).max() <= 1e-5 * numpy.abs(ref).max()
    back = _execute_split(ctx, shape, numpy.float32, batch, got[0], got[1], inverse=True, expect="fused2")
    x = re + 1j * im
    assert numpy.abs((back[0] + 1j * back[1]) - x).sum() / numpy.abs(x).sum() < 1.1e-6
    monkeypatch.setenv("PYFFT_AMD_STRATEGY", "chain")
    want = _execute_split(ctx, shape, numpy.float32, batch, re, im, expect="chain")
    assert numpy.abs((want[0] - got[0]) + 1j * (want[1] - got[1])).sum() / numpy.abs(want[0] + 1j * want[1]).sum() < 5e-7
    monkeypatch.setenv("PYFFT_AMD_STRATEGY", "auto")
    monkeypatch.setenv("PYFFT_AMD_NO_SPLIT_ROWFIRST", "1")
    assert ctx.getPlan(shape, dtype=numpy.float32).strategy(batch)[0] == "pipelined"


# ---- split-complex fp64 (float64 planes) on the persistent kernels, planes streamed non-temporally -----------------------------------
SPLIT_FP64_CASES = [((1 << 16,), 270), ((1 << 17,), 131), ((1 << 18,), 67), ((1 << 19,), 35), ((1 << 20,), 29), ((1024, 1024), 30)]


@pytest.mark.parametrize("shape,batch", SPLIT_FP64_CASES, ids=str)
def test_fused_split_planes_fp64(ctx, monkeypatch, shape, batch):
    """float64 planes beyond the chain threshold: 1-D 2^16 ... 2^20 and the published 1024 x 1024 on the persistent kernels (16 columns
    of an fp64 plane are a whole 128-byte line; the tiles stream the planes with non-temporal loads and stores, second batch of
    round 4).  numpy with the reference's fp64 thresholds on sampled transforms, planes untouched, in place == out of place, inverse
    round trip, and the chain's result to rounding."""
    n = int(numpy.prod(shape))
    rng = numpy.random.default_rng(1800 + n % 89 + len(shape))
    re = _noise(rng, n * batch, numpy.float64)
    im = _noise(rng, n * batch, numpy.float64)
    monkeypatch.setenv("PYFFT_AMD_STRATEGY", "auto")
    got = _execute_split(ctx, shape, numpy.float64, batch, re, im, expect="fused2")
    inp = _execute_split(ctx, shape, numpy.float64, batch, re, im, inplace=True, expect="fused2")
    assert numpy.array_equal(inp[0], got[0]) and numpy.array_equal(inp[1], got[1])
    for item in (0, batch // 2, batch - 1):
        sl = slice(item * n, (item + 1) * n)
        ref = numpy.fft.fftn((re[sl] + 1j * im[sl]).reshape(shape)).reshape(-1)
        g = got[0][sl] + 1j * got[1][sl]
        assert numpy.abs(ref - g).sum() / numpy.abs(ref).sum() < 1e-11
        assert numpy.abs(ref - g).max() <= 1e-10 * numpy.abs(ref).max()
    back = _execute_split(ctx, shape, numpy.float64, batch, got[0], got[1], inverse=True, expect="fused2")
    x = re + 1j * im
    assert numpy.abs((back[0] + 1j * back[1]) - x).sum() / numpy.abs(x).sum() < 1e-11
    monkeypatch.setenv("PYFFT_AMD_STRATEGY", "chain")
    want = _execute_split(ctx, shape, numpy.float64, batch, re, im, expect="chain")
    assert numpy.abs((want[0] - got[0]) + 1j * (want[1] - got[1])).sum() / numpy.abs(want[0] + 1j * want[1]).sum() < 1e-14

"""GPU tests of the persistent launches (csrc/fft_fused2.hpp, fft_fusedp.hpp, fft_fused2r.hpp, fft_fusedx_f64.hip): both passes -- or both
pass pairs -- of every transform of a batch in ONE launch, the intermediate in a ring that stays in the last-level cache.  Every shape class
of the planner's tuning table (pyfft_amd/tuning_gfx950.json) in both precisions and layouts, the counter sets and error word the launches
synchronise through, and BASELINE.json configs[4] as stated on one GPU.  Whole arrays against the plain chain (same tile code: same bits)
where that holds, sampled transforms against numpy.fft with the reference's thresholds (test/test_errors.py:20-23)."""
import ctypes
import os

import numpy
import pytest

import pyfft_oracle as oracle
from helpers import EPS_F, MAX_F, _execute, _execute_split, _noise, _test_data

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_async_error_mailbox(ctx):
    """Asynchronous executes of the fused kernel report through the plan's pinned error word (round 4: the kernel writes host
    memory itself; rounds 2-3 copied a device word back behind every launch): check() never blocks, finish() synchronises, a
    non-zero word raises once at the next host-visible point."""
    n, batch = 1 << 20, 64
    stream = ctx.hip.Stream()
    plan = ctx.getPlan((n,), dtype=numpy.complex64, stream=stream)
    a = ctx.toGpu(numpy.ones(n * batch, dtype=numpy.complex64))
    b = ctx.allocate((n * batch,), numpy.complex64)
    assert plan.strategy(batch)[0] == "fused2"
    for _ in range(5):
        assert plan.execute(a, b, batch=batch) is stream
    plan.check()
    assert plan._errword is not None and plan._mailbox is None
    plan.finish()
    res = b.get().reshape(batch, n)
    assert numpy.allclose(res[:, 0], n) and numpy.abs(res[:, 1:]).max() < 1e-2
    # what a timed-out persistent launch leaves behind: the word is non-zero -> raised once, and the counters are re-zeroed
    plan._errword._word.value = 1
    with pytest.raises(RuntimeError, match="time-out"):
        plan.check()
    plan.check()
    assert plan._counters_clean is False
    plan._errword._word.value = 1
    with pytest.raises(RuntimeError, match="time-out"):
        plan.execute(a, b, batch=batch)
    N = ctx.hip.N
    N.check(N.lib.mifft_memset(b.ptr, 0, b.nbytes, None), "memset")
    plan.execute(a, b, batch=batch, wait_for_finish=True)                   # and the plan works again
    res = b.get().reshape(batch, n)
    assert numpy.allclose(res[:, 0], n) and numpy.abs(res[:, 1:]).max() < 1e-2
    # the copy-back mailbox (development strategy xcd2) still turns a non-zero word into the same error
    plan._handle_errors([])
    with pytest.raises(RuntimeError):
        plan._handle_errors([("fused2", 1)])


@pytest.mark.parametrize("dtype", [numpy.complex64, numpy.float32], ids=["interleaved", "split"])
def test_xcd2_strategy_bit_identical_and_in_place(ctx, dtype, monkeypatch):
    """The XCD-cooperative kernel (csrc/fft_xcd2.hpp, development strategy 'xcd'): bit-identical to the chain strategy,
    forward and inverse, out of place and in place (it needs no temp buffer), and its error word stays clear.  The kernel is part
    of `make DEV=1` builds of the library only (mifft_has_feature)."""
    from pyfft_amd import _native as N
    if N.lib.mifft_has_feature(N.FEATURE_XCD2) != 1:
        pytest.skip("development strategy: not in the default build of libmifft.so (make DEV=1)")
    n, batch = 1 << 20, 72
    split = numpy.dtype(dtype).kind == "f"
    rng = numpy.random.default_rng(77)
    re = rng.standard_normal(n * batch).astype(numpy.float32)
    im = rng.standard_normal(n * batch).astype(numpy.float32)
    host = [re, im] if split else [(re + 1j * im).astype(numpy.complex64)]

    def run(strategy, inverse, inplace):
        monkeypatch.setenv("PYFFT_AMD_STRATEGY", strategy)
        plan = ctx.getPlan((n,), dtype=dtype, context=ctx.context)
        assert plan.strategy(batch)[0] == ("xcd2" if strategy == "xcd" else "chain")
        src = [ctx.toGpu(h) for h in host]
        dst = src if inplace else [ctx.allocate(h.shape, h.dtype) for h in host]
        if inplace:
            plan.execute(*src, inverse=inverse, batch=batch)
        else:
            plan.execute(*(src + dst), inverse=inverse, batch=batch)
            for s_, h in zip(src, host):
                assert numpy.array_equal(s_.get(), h)
        return [d.get() for d in dst]

    for inverse in (False, True):
        want = run("chain", inverse, False)
        for inplace in (False, True):
            got = run("xcd", inverse, inplace)
            for w, g in zip(want, got):
                assert numpy.array_equal(w.view(numpy.uint32), g.view(numpy.uint32))
    ref = numpy.fft.fft((re[:n].astype(numpy.float64) + 1j * im[:n]))
    fw = run("xcd", False, False)
    got = (fw[0][:n] + 1j * fw[1][:n]) if split else fw[0][:n]
    assert numpy.abs(got - ref).max() <= 1e-5 * numpy.abs(ref).max()


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


# the driver's GPU step has a time limit: the cases that repeat a kernel family on one more shape run with the soak switch PYFFT_AMD_SWEEP
_SOAK = bool(os.environ.get("PYFFT_AMD_SWEEP"))


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
    slower than the two launches (docs/negative_results.md): part of `make DEV=1` builds of the library only."""
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
                         ((64, 128, 64), 115), ((64, 64, 128), 117), ((128, 64, 64), 113),
                         ((32, 32, 128), 300),        # round 6: a shape of the reference's own benchmark list, y = 8 x 4 ...
                         ((32, 64, 128), 151), ((32, 128, 128), 75), ((64, 32, 128), 149), ((128, 32, 128), 77), ((32, 64, 64), 301),
                         ((32, 128, 64), 153)]        # ... and its neighbours with 32-point axes on the same tile kinds


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


N5 = 1 << 22           # configuration 5: 1-D c2c fp32 N = 2^22


CHUNK5 = 256           # one resident chunk: 8 GiB in + 8 GiB out


SHARE5 = 8192          # per GPU: 65536 transforms over 8 GPUs (BASELINE.json configs[4]; kernel.py:99-121: batch -> grid)


BLK5 = 16              # the synthetic data set is periodic: 16 seeded transforms (512 MiB)


_c5_cache = {}


def _c5_block():
    """The seeded block of configuration 5 and numpy's transform of every item of it (complex128), computed once per session."""
    if "block" not in _c5_cache:
        rng = numpy.random.default_rng(1005)
        re = rng.standard_normal((BLK5, N5)).astype(numpy.float32)
        im = rng.standard_normal((BLK5, N5)).astype(numpy.float32)
        block = numpy.empty((BLK5, N5), numpy.complex64)
        block.real = re
        block.imag = im
        _c5_cache["block"] = block
        _c5_cache["refs"] = [numpy.fft.fft(block[i].astype(numpy.complex128)) for i in range(BLK5)]
    return _c5_cache["block"], _c5_cache["refs"]


def _fetch(N, ptr, item, size=N5):
    out = numpy.empty(size, numpy.complex64)
    N.check(N.lib.mifft_memcpy_d2h(out.ctypes.data, ptr + item * size * 8, out.nbytes, None))
    return out


def _close(got, ref):
    got = got.astype(numpy.complex128)
    return (numpy.abs(ref - got).sum() / numpy.abs(ref).sum() < EPS_F) and (numpy.abs(ref - got).max() <= MAX_F * numpy.abs(ref).max())


def test_config5_per_gpu_share(ctx):
    """The full per-GPU share of configuration 5 as the survey's streaming loop: 32 chunks of 256 transforms through ONE plan and
    ONE pair of 8 GiB buffers.  Global transform g holds block item (g + g // 256) % 16 -- every chunk is the block rotated one item
    further, so a chunk that was skipped, executed on stale data or mixed up with its neighbour cannot pass -- and the input buffer
    is refilled between chunks from a device copy of the block.  First / middle / last transform of EVERY chunk against numpy."""
    from pyfft_amd import _native as N
    hip = ctx.hip
    block, refs = _c5_block()
    item_bytes = N5 * 8
    dev_block = hip.to_gpu(block.reshape(-1))
    a = hip.DeviceArray((CHUNK5 * N5,), numpy.complex64)
    b = hip.DeviceArray((CHUNK5 * N5,), numpy.complex64)
    plan = hip.Plan(N5, dtype=numpy.complex64)
    assert plan.strategy(CHUNK5)[0] == "fused2"             # the persistent kernel of the stated configuration

    def refill(c):
        """a[s] <- block[(s + c) % 16] for the 256 transforms s of chunk c"""
        rot = c % BLK5
        head = (BLK5 - rot) * item_bytes
        N.check(N.lib.mifft_memcpy_d2d(a.ptr, dev_block.ptr + rot * item_bytes, head, None))
        if rot:
            N.check(N.lib.mifft_memcpy_d2d(a.ptr + head, dev_block.ptr, rot * item_bytes, None))
        done = BLK5 * item_bytes
        while done < a.nbytes:
            n = min(done, a.nbytes - done)
            N.check(N.lib.mifft_memcpy_d2d(a.ptr + done, a.ptr, n, None))
            done += n

    chunks = SHARE5 // CHUNK5
    checked = 0
    for c in range(chunks):
        refill(c)
        plan.execute(a, b, batch=CHUNK5)
        for s in (0, CHUNK5 // 2 + 1, CHUNK5 - 1):
            g = c * CHUNK5 + s
            assert _close(_fetch(N, b.ptr, s), refs[(g + g // CHUNK5) % BLK5]), ("chunk", c, "transform", g)
            checked += 1
        if c in (0, chunks // 2, chunks - 1):               # the input of an out-of-place execute stays what it was
            assert numpy.array_equal(_fetch(N, a.ptr, CHUNK5 - 1), block[(CHUNK5 - 1 + c) % BLK5])
    assert checked == 3 * chunks
    # the last chunk's result, inverse in place: the round trip
    plan.execute(b, batch=CHUNK5, inverse=True)
    for s in (0, CHUNK5 - 1):
        want = block[(s + chunks - 1) % BLK5].astype(numpy.complex128)
        got = _fetch(N, b.ptr, s).astype(numpy.complex128)
        assert numpy.abs(want - got).sum() / numpy.abs(want).sum() < EPS_F


def test_config5_share_as_one_execute(ctx):
    """The same share as ONE in-place execute(batch = 8192): 256 GiB resident (byte offsets up to 2^38, 2^35 elements, 16393 counters
    per set), skipped with a message where the allocation is refused.  Sampled transforms against numpy, periodic input ->
    bit-identical outputs across the whole buffer, inverse in place -> the input."""
    from pyfft_amd import _native as N
    hip = ctx.hip
    block, refs = _c5_block()
    try:
        buf = hip.DeviceArray((SHARE5 * N5,), numpy.complex64)
    except RuntimeError as e:
        pytest.skip("mifft_malloc refuses 256 GiB on this device: %s" % (str(e)[:200],))
    hb = block.reshape(-1).view(numpy.uint8)
    N.check(N.lib.mifft_memcpy_h2d(buf.ptr, hb.ctypes.data, hb.nbytes, None))
    done = hb.nbytes
    while done < buf.nbytes:
        n = min(done, buf.nbytes - done)
        N.check(N.lib.mifft_memcpy_d2d(buf.ptr + done, buf.ptr, n, None))
        done += n
    N.check(N.lib.mifft_device_sync())
    plan = hip.Plan(N5, dtype=numpy.complex64)
    assert plan.strategy(SHARE5)[0] == "fused2"
    plan.execute(buf, batch=SHARE5)
    samples = [0, 1, BLK5 - 1, BLK5, SHARE5 // 2 - 1, SHARE5 // 2, SHARE5 // 2 + 5, SHARE5 - BLK5 - 3, SHARE5 - 2, SHARE5 - 1]
    first = {}
    for g in samples:
        got = _fetch(N, buf.ptr, g)
        assert _close(got, refs[g % BLK5]), g
        # periodic input -> bit-identical output wherever the item lies in the 256 GiB
        if g % BLK5 in first:
            assert numpy.array_equal(first[g % BLK5].view(numpy.uint32), got.view(numpy.uint32)), g
        else:
            first[g % BLK5] = got
    plan.execute(buf, batch=SHARE5, inverse=True)
    for g in (0, SHARE5 // 2 + 5, SHARE5 - 1):
        want = block[g % BLK5].astype(numpy.complex128)
        got = _fetch(N, buf.ptr, g).astype(numpy.complex128)
        assert numpy.abs(want - got).sum() / numpy.abs(want).sum() < EPS_F, g
    plan.close()
    del buf


def test_config5_share_every_output_bit_identical_to_its_period_mate(ctx):
    """The whole-array check of the stated configuration (the share tests sampled ten of 8192 transforms): the 256 GiB
    data set is periodic with period 16 transforms, so EVERY one of the 8192 outputs of one in-place execute must carry the bits of the
    output 16 transforms before it -- compared on the device (mifft_aux_count_mismatch: the buffer against itself shifted by one period,
    255.5 GiB of 16-byte words), which a fault confined to any ring position, ticket range or address bit cannot survive; then the 16
    distinct outputs against numpy, and the same after the inverse in place."""
    import ctypes
    from pyfft_amd import _native as N
    hip = ctx.hip
    block, refs = _c5_block()
    try:
        buf = hip.DeviceArray((SHARE5 * N5,), numpy.complex64)
    except RuntimeError as e:
        pytest.skip("mifft_malloc refuses 256 GiB on this device: %s" % (str(e)[:200],))
    hb = block.reshape(-1).view(numpy.uint8)
    N.check(N.lib.mifft_memcpy_h2d(buf.ptr, hb.ctypes.data, hb.nbytes, None))
    done = hb.nbytes
    while done < buf.nbytes:
        n = min(done, buf.nbytes - done)
        N.check(N.lib.mifft_memcpy_d2d(buf.ptr + done, buf.ptr, n, None))
        done += n
    N.check(N.lib.mifft_device_sync())
    word = ctypes.c_void_p()
    N.check(N.lib.mifft_host_alloc(ctypes.byref(word), 64), "mifft_host_alloc")
    count = ctypes.c_uint64.from_address(word.value)
    period = BLK5 * N5 * 8

    def mismatches():
        count.value = 0
        N.check(N.lib.mifft_aux_count_mismatch(buf.ptr, buf.ptr + period, buf.nbytes - period, word.value, None), "mifft_aux_count_mismatch")
        N.check(N.lib.mifft_device_sync())
        return int(count.value)

    try:
        assert mismatches() == 0, "the periodic input is not periodic"
        # (the checker itself: one flipped word in the middle of the buffer is two mismatches -- with the item before and the item after)
        probe = numpy.zeros(4, numpy.uint32)
        mid = buf.ptr + (SHARE5 // 2 + 3) * N5 * 8 + 4096
        N.check(N.lib.mifft_memcpy_d2h(probe.ctypes.data, mid, 16, None))
        flipped = probe ^ numpy.uint32(1)
        N.check(N.lib.mifft_memcpy_h2d(mid, flipped.ctypes.data, 16, None))
        assert mismatches() == 2
        N.check(N.lib.mifft_memcpy_h2d(mid, probe.ctypes.data, 16, None))
        plan = hip.Plan(N5, dtype=numpy.complex64)
        assert plan.strategy(SHARE5)[0] == "fused2"
        plan.execute(buf, batch=SHARE5)
        assert mismatches() == 0, "some transform of the share differs from its period-mate"
        for g in range(BLK5):
            assert _close(_fetch(N, buf.ptr, g), refs[g]), g
        plan.execute(buf, batch=SHARE5, inverse=True)
        assert mismatches() == 0, "inverse: some transform of the share differs from its period-mate"
        for g in (0, BLK5 - 1):
            want = block[g].astype(numpy.complex128)
            got = _fetch(N, buf.ptr, g).astype(numpy.complex128)
            assert numpy.abs(want - got).sum() / numpy.abs(want).sum() < EPS_F, g
        plan.close()
    finally:
        N.lib.mifft_device_sync()
        N.lib.mifft_host_free(word.value)
        del buf


def test_direct_abi_two_set_launch_refuses_capture_and_null_error_word(ctx):
    """C-ABI users of mifft_launch_fused2: the two-set form on a capturing stream is MIFFT_E_INVALID (a replay would start on dirty
    counters), and so is the two-set form without an error word of its own (the next launch would zero the default one)."""
    import ctypes
    from pyfft_amd import _native as N
    hip = ctx.hip
    n, batch = 1 << 18, 160
    s = hip.Stream()
    plan = hip.Plan(n, dtype=numpy.complex64, stream=s)
    a = hip.DeviceArray((n * batch,), numpy.complex64)
    N.check(N.lib.mifft_memset(a.ptr, 0, a.nbytes, s.handle))
    plan.execute(a, batch=batch)
    s.synchronize()
    strat = plan.strategy(batch)
    assert strat[0] == "fused2"
    _, lag, ring, grid = strat
    descs = plan._descriptors(batch, True, False)
    base = plan._context.pointer_of(plan._counters)
    nb = plan._counter_bytes
    N.check(N.lib.mifft_memset(base, 0, 3 * nb, s.handle))
    s.synchronize()
    plan._counters_clean, plan._counter_set = True, 0
    tmp = plan._context.pointer_of(plan._tempmemobj)

    def launch(sync):
        return N.lib.mifft_launch_fused2(ctypes.byref(descs[0]), ctypes.byref(descs[1]), a.ptr, None, a.ptr, None, tmp, None, ring, lag,
                                         ctypes.byref(sync), grid, s.handle)

    assert launch(N.MifftFusedSync(base, base + nb, None)) == N.E_INVALID
    assert b"error word" in N.lib.mifft_last_error()
    N.check(N.lib.mifft_stream_begin_capture(s.handle))
    rc = launch(N.MifftFusedSync(base, base + nb, plan._errword.ptr))
    msg = N.lib.mifft_last_error()
    rc_single = launch(N.MifftFusedSync(base + 2 * nb, None, plan._errword.ptr))
    h = ctypes.c_void_p()
    N.check(N.lib.mifft_stream_end_capture(s.handle, ctypes.byref(h)))
    N.lib.mifft_graph_destroy(h)
    assert rc == N.E_INVALID and b"capturing" in msg
    assert rc_single == 0
    s.synchronize()

# ---- round 6: the (y, x) planes of a 3-D transform bigger than a pipeline chunk on the persistent 2-D kernels ------------------------
PLANE_FUSED_CASES = [((128, 1024, 256), numpy.complex64, 2), ((32, 2048, 512), numpy.complex64, 2), ((256, 512, 256), numpy.complex64, 2),
                     ((128, 512, 512), numpy.complex64, 2), ((64, 1024, 512), numpy.complex64, 2), ((128, 512, 256), numpy.complex128, 2),
                     ((32, 1024, 512), numpy.complex128, 2), ((16, 1024, 1024), numpy.float64, 2), ((512, 512, 512), numpy.complex64, 1),
                     ((256, 512, 512), numpy.complex64, 1)]


@pytest.mark.parametrize("shape,dtype,batch", PLANE_FUSED_CASES, ids=str)
def test_plane_fused_3d(ctx, monkeypatch, shape, dtype, batch):
    """3-D transforms of more than half the last-level cache whose chain is ROW x + COL y + COL z (no pass pair): the x and y passes run as ONE persistent
    launch over the batch * nz (y, x) planes -- the 2-D plan's kernel and ring --, the z pass as the chain's plain launch (strategy
    "fused2z"; rounds 1-5: the leading passes slab by slab through the pipelined launcher, or three plain launches for small batches;
    pyfft/plan.py:135-171: the reference's chain is x kernels, then y, then z).  One case per 2-D rule of the tuning table that such a
    shape can match, and 512^3 / (256, 512, 512) with batch 1 (their z passes run on 32-column tiles: rows 2 MiB apart).  First / last transform against numpy (reference thresholds), in place == out of place,
    input untouched, the inverse round trip, and the whole result against the plan without the route (other kernels for x and y:
    rounding-level agreement)."""
    cdt = numpy.dtype(dtype)
    split = cdt.kind == "f"
    double = cdt in (numpy.dtype(numpy.complex128), numpy.dtype(numpy.float64))
    eps, mx = (1e-11, 1e-10) if double else (EPS_F, MAX_F)
    ctype = numpy.complex128 if double else numpy.complex64
    size = int(numpy.prod(shape))
    data = _test_data(shape, ctype, batch, 6600 + shape[0] + shape[1] // 256)
    plan = ctx.getPlan(shape, dtype=dtype)
    assert plan.strategy(batch)[0] == "fused2z", plan.strategy(batch)
    assert len(plan.pass_list()) == 3

    def run(inplace=False, inverse=False, src=None, expect="fused2z"):
        d = data if src is None else src
        if split:
            re, im = _execute_split(ctx, shape, dtype, batch, numpy.ascontiguousarray(d.real), numpy.ascontiguousarray(d.imag),
                                    inplace=inplace, inverse=inverse, expect=expect)
            return (re + 1j * im).astype(ctype)
        return _execute(ctx, shape, dtype, batch, d, inplace=inplace, inverse=inverse, expect=expect)
    got = run()
    assert numpy.array_equal(run(inplace=True), got)
    flat, gflat = data.reshape(-1), got.reshape(-1)
    for item in sorted({0, batch - 1}):
        ref = numpy.fft.fftn(flat[item * size:(item + 1) * size].reshape(shape).astype(numpy.complex128)).reshape(-1)
        d = gflat[item * size:(item + 1) * size]
        assert numpy.abs(ref - d).sum() / numpy.abs(ref).sum() < eps
        assert numpy.abs(ref - d).max() <= mx * numpy.abs(ref).max()
        del ref
    back = run(inverse=True, src=got)
    assert oracle.difference(data, back, batch) < eps
    del back
    monkeypatch.setenv("PYFFT_AMD_NO_PLANE_FUSED", "1")
    other = ctx.getPlan(shape, dtype=dtype).strategy(batch)[0]
    assert other in ("chain", "pipelined")
    want = run(expect=other)
    assert oracle.difference(want, got, batch) < (1e-14 if double else 5e-7)

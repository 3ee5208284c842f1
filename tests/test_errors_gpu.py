"""GPU parity tests: the reference's accuracy protocol (test/test_errors.py:18-114) run against
the HIP path through the C ABI, with numpy.fft on the complex128-upcast input as the reference
answer and the CPU oracle's restatement of the reference chain as a second checker.

Tolerances are the reference's own: L1-relative `difference` < 1.1e-6 (fp32) / 1e-11 (fp64)
(test_errors.py:20-23), plus the north star's max-norm bound max|out-ref| <= 1e-5 * max|ref|.
"""
import json
import os

import numpy
import pytest

import pyfft_oracle as oracle
from helpers import COMPLEX_DTYPES, DOUBLE_DTYPES, getDimensions

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_protocol(ctx, shape, dtype, batch, seed=1234, fast_math=True, check_oracle=True):
    epsilon = 1e-11 if dtype in DOUBLE_DTYPES else 1.1e-6
    split = dtype not in COMPLEX_DTYPES
    complex_dtype = numpy.complex128 if dtype in DOUBLE_DTYPES else numpy.complex64

    if split:
        data_re, data_im = oracle.get_test_data(shape, dtype, batch, seed)
        data = (data_re + 1j * data_im).astype(complex_dtype)
    else:
        data = oracle.get_test_data(shape, dtype, batch, seed)

    numpy_fw = oracle.numpy_fft(numpy.fft.fftn, data, batch)
    if data.size <= (1 << 22):
        numpy_res = oracle.numpy_fft(numpy.fft.ifftn, numpy_fw, batch)
        numpy_err = oracle.difference(numpy_res, data, batch)
    else:
        # (the reference's protocol also round-trips numpy itself, test/test_errors.py:41-43: a check of the checker, skipped for the
        # 2^23 ... 2^24-point cases of round 6, where it is seconds of host time per case)
        numpy_err = 0.0

    if split:
        a_re, a_im = ctx.toGpu(data_re), ctx.toGpu(data_im)
        b_re, b_im = ctx.allocate(data_re.shape, data_re.dtype), ctx.allocate(data_im.shape, data_im.dtype)
    else:
        a_gpu = ctx.toGpu(data)
        b_gpu = ctx.allocate(data.shape, data.dtype)

    plan = ctx.getPlan(shape, dtype=dtype, context=ctx.context, normalize=True,
                       wait_for_finish=True, fast_math=fast_math)

    def fetch(re, im=None):
        if split:
            return ctx.fromGpu(re, data_re.shape, data_re.dtype) + 1j * ctx.fromGpu(im, data_im.shape, data_im.dtype)
        return ctx.fromGpu(re, data.shape, data.dtype)

    # out of place forward; the input must not be modified (plan.py:200-248 contract)
    if split:
        plan.execute(a_re, a_im, b_re, b_im, batch=batch)
        fw_outplace = fetch(b_re, b_im)
        assert numpy.array_equal(fetch(a_re, a_im), data), "out-of-place execute modified its input"
        plan.execute(b_re, b_im, a_re, a_im, batch=batch, inverse=True)
        res_outplace = fetch(a_re, a_im)
    else:
        plan.execute(a_gpu, b_gpu, batch=batch)
        fw_outplace = fetch(b_gpu)
        assert numpy.array_equal(fetch(a_gpu), data), "out-of-place execute modified its input"
        plan.execute(b_gpu, a_gpu, batch=batch, inverse=True)
        res_outplace = fetch(a_gpu)
    err_outplace = oracle.difference(res_outplace, data, batch)

    # in place forward / inverse
    if split:
        a_re, a_im = ctx.toGpu(data_re), ctx.toGpu(data_im)
        plan.execute(a_re, a_im, batch=batch)
        fw_inplace = fetch(a_re, a_im)
        plan.execute(a_re, a_im, batch=batch, inverse=True)
        res_inplace = fetch(a_re, a_im)
    else:
        a_gpu = ctx.toGpu(data)
        plan.execute(a_gpu, batch=batch)
        fw_inplace = fetch(a_gpu)
        plan.execute(a_gpu, batch=batch, inverse=True)
        res_inplace = fetch(a_gpu)
    err_inplace = oracle.difference(res_inplace, data, batch)

    err_inout_fw = oracle.difference(fw_inplace, fw_outplace, batch)
    err_inout_res = oracle.difference(res_inplace, res_outplace, batch)
    diff_err = oracle.difference(numpy_fw, fw_inplace, batch)
    max_err = numpy.abs(fw_inplace - numpy_fw).max() / numpy.abs(numpy_fw).max()

    assert err_inout_fw < epsilon, "inplace-outplace intermediate error: %g" % err_inout_fw
    assert err_inout_res < epsilon, "inplace-outplace final error: %g" % err_inout_res
    assert numpy_err < epsilon, "numpy forward-inverse error: %g" % numpy_err
    assert err_inplace < epsilon, "forward-inverse inplace error: %g" % err_inplace
    assert err_outplace < epsilon, "forward-inverse outplace error: %g" % err_outplace
    assert diff_err < epsilon, "difference between HIP path and numpy: %g" % diff_err
    assert max_err <= (1e-5 if dtype not in DOUBLE_DTYPES else 1e-10), "max-norm error %g" % max_err

    if check_oracle:
        # the CPU restatement of the reference's own kernel chain, same inputs
        ora_fw = oracle.execute(data, shape, dtype=dtype, batch=batch)
        assert oracle.difference(ora_fw.astype(numpy.complex128), fw_inplace, batch) < 2 * epsilon
    return err_inplace, diff_err


# one representative of every row of SURVEY.md Appendix B: single launch, 2-pass, 3-pass 1-D,
# short and long strided axes, with and without temp, odd and even chain lengths
SHAPES_1D = [(2,), (4,), (8,), (16,), (32,), (64,), (128,), (256,), (512,), (1024,), (2048,), (4096,),
             (8192,), (1 << 14,), (1 << 16,), (1 << 17,)]
SHAPES_2D = [(16, 16), (128, 32), (64, 256), (2, 2), (4, 1024), (1024, 4), (256, 256), (2048, 8), (8, 8192)]
SHAPES_3D = [(16, 16, 16), (8, 8, 64), (32, 16, 8), (2, 4, 2), (128, 16, 16), (16, 128, 4), (4, 4, 2048)]
ALL_DTYPES = [numpy.complex64, numpy.float32, numpy.complex128, numpy.float64]


@pytest.mark.parametrize("dtype", ALL_DTYPES, ids=lambda d: numpy.dtype(d).name)
@pytest.mark.parametrize("shape", SHAPES_1D + SHAPES_2D + SHAPES_3D, ids=str)
def test_errors_batch_1_and_3(ctx, shape, dtype):
    x, y, z = getDimensions(shape)
    for batch in (1, 3):
        if x * y * z * batch > (1 << 19):
            continue
        run_protocol(ctx, shape, dtype, batch, seed=1000 + batch)


@pytest.mark.parametrize("dtype", [numpy.complex64, numpy.float64], ids=lambda d: numpy.dtype(d).name)
@pytest.mark.parametrize("shape,batch", [((8,), 4096), ((16,), 1024), ((256,), 128), ((1024,), 16), ((16, 16), 128),
                                         ((16, 16, 16), 16), ((128, 128), 16), ((8192,), 16), ((13,), 0)][:-1], ids=str)
def test_errors_batched(ctx, shape, batch, dtype):
    """Batch sizes of the reference sweep (test_errors.py:139) incl. ragged tails of tiles."""
    run_protocol(ctx, shape, dtype, batch, seed=77)
    run_protocol(ctx, shape, dtype, batch - 1 if batch > 1 else 5, seed=78, check_oracle=False)


@pytest.mark.parametrize("dtype,n", [(numpy.complex64, 1 << k) for k in range(8, 16)] +
                         [(numpy.complex128, 1 << k) for k in range(10, 15)] +
                         [(numpy.float32, 1 << 15), (numpy.float64, 1 << 14)],      # (round 6: the longest rows on split-complex planes)
                         ids=lambda v: str(v) if isinstance(v, int) else numpy.dtype(v).name)
def test_register_edged_rows(ctx, dtype, n):
    """Every register-edged ROW kernel (csrc/fft_row2.hpp: several rows per work-group, plain and half-exchange
    forms) with a batch that leaves a ragged last tile, and as the first pass of a 2-D plan."""
    run_protocol(ctx, (n,), dtype, 37, seed=n + 5)
    if n <= 4096:
        run_protocol(ctx, (128, n), dtype, 3, seed=n + 6)  # numpy shape (y, x): ROW over x = n, then a strided pass


def _fixed_nd_shapes(prec):
    """(x, y, z) shapes of the generated tables csrc/fft_nd2_f32_*.hip / fft_nd2_f64_*.hip (tools/gen_nd2_tables.py is
    their single source)."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "gen_nd2_tables.py")
    spec = importlib.util.spec_from_file_location("gen_nd2_tables", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.shapes(prec)


FIXED_ND_F32 = _fixed_nd_shapes("f32")
FIXED_ND_F64 = _fixed_nd_shapes("f64")


@pytest.mark.parametrize("dtype,xyz", [(numpy.complex64, s) for s in FIXED_ND_F32] + [(numpy.complex128, s) for s in FIXED_ND_F64],
                         ids=lambda v: "x".join(map(str, v)) if isinstance(v, tuple) else numpy.dtype(v).name)
def test_fixed_shape_nd_kernels(ctx, dtype, xyz):
    """Every fixed-shape N-D kernel (csrc/fft_nd2_*.hip) with a batch that leaves a ragged last tile."""
    x, y, z = xyz
    shape = (y, x) if z == 1 else (z, y, x)  # numpy order, x last
    n = x * y * z
    tile = 4096 if dtype == numpy.complex64 else 2048
    batch = 3 if n >= tile else (tile // n) * 2 + 1
    run_protocol(ctx, shape, dtype, batch, seed=n + batch, check_oracle=(n <= 4096 and n * batch <= 16384))


@pytest.mark.parametrize("dtype", [numpy.complex64, numpy.float32, numpy.complex128], ids=lambda d: numpy.dtype(d).name)
@pytest.mark.parametrize("shape,batch", [((32, 32, 128), 2), ((16, 16, 128), 3), ((64, 64, 64), 2), ((16, 16), 37),
                                         ((32, 32, 32), 3), ((128, 128), 3), ((128, 128, 128), 1)], ids=str)
def test_fixed_shape_nd_planes(ctx, shape, batch, dtype):
    """3-D shapes whose (y, x) planes run on the fixed-shape kernels before the strided z pass; float32 (split planes)
    takes the run-time-shaped kernel."""
    run_protocol(ctx, shape, dtype, batch, seed=31 + batch)


@pytest.mark.parametrize("dtype", [numpy.complex128, numpy.float64], ids=lambda d: numpy.dtype(d).name)
@pytest.mark.parametrize("shape,batch", [((1 << 18,), 3), ((1 << 17,), 5), ((512, 64), 9), ((512, 16, 2), 4)], ids=str)
def test_fp64_two_phase_col_512(ctx, shape, batch, dtype):
    """fp64 two-phase COL kernels at L = 512 (csrc/fft_col2_f64.hip: single exchange buffer): transposing first pass
    with inter-pass twiddles, plain last pass, and strided y / z axes."""
    run_protocol(ctx, shape, dtype, batch, seed=77 + batch)


@pytest.mark.parametrize("dtype", [numpy.complex64, numpy.float64], ids=lambda d: numpy.dtype(d).name)
@pytest.mark.parametrize("shape", [(16, 1), (1, 16), (1, 1, 16), (8, 1, 8), (1, 64, 1), (4096, 1), (1, 2048, 4), (2, 1, 1)], ids=str)
def test_unit_axes(ctx, shape, dtype):
    """Axes of length 1 (the reference lists y == 1 / z == 1 plans as a TODO, TODO.txt:6-8; plan.py:149,160,164 skip
    them): every placement of the unit axes gives the transform over the remaining ones."""
    run_protocol(ctx, shape, dtype, 3, seed=5 + len(shape))


@pytest.mark.parametrize("dtype", [numpy.complex64, numpy.float32], ids=lambda d: numpy.dtype(d).name)
def test_errors_large_1d(ctx, dtype):
    """N = 2^20 (BASELINE config 2 shape, small batch): the reference's largest 1-D test size
    (test_errors.py:125-126)."""
    run_protocol(ctx, (1 << 20,), dtype, 2, seed=1002, check_oracle=False)


def test_errors_config3_shape(ctx):
    """1024 x 1024 c64 (BASELINE config 3 shape) at batch 2."""
    run_protocol(ctx, (1024, 1024), numpy.complex64, 2, seed=1003, check_oracle=False)


@pytest.mark.parametrize("dtype", [numpy.complex128, numpy.float64], ids=lambda d: numpy.dtype(d).name)
def test_errors_config4_shape_reduced(ctx, dtype):
    """3-D fp64 in both layouts (BASELINE config 4 at 64^3 and 256x16x16)."""
    run_protocol(ctx, (64, 64, 64), dtype, 1, seed=1004, check_oracle=False)
    run_protocol(ctx, (256, 16, 16), dtype, 2, seed=1004, check_oracle=False)


# ---- execution strategies: the fused persistent kernel and the stream-pipelined chunks must give exactly the
# ---- bits of the one-launch-per-pass chain (same kernels' arithmetic), and the reference protocol must hold
def _run_strategy(ctx, monkeypatch, strat, shape, batch, data, inplace=False, inverse=False):
    monkeypatch.setenv("PYFFT_AMD_STRATEGY", strat)
    plan = ctx.getPlan(shape, dtype=numpy.complex64)
    assert plan.strategy(batch)[0] == {"fused": "fused2", "chain": "chain", "pipelined": "pipelined"}[strat]
    a = ctx.toGpu(data)
    if inplace:
        plan.execute(a, batch=batch, inverse=inverse)
        return a.get()
    b = ctx.allocate(data.shape, data.dtype)
    plan.execute(a, b, batch=batch, inverse=inverse)
    assert numpy.array_equal(a.get(), data)
    return b.get()


@pytest.mark.parametrize("n,batch", [(1 << 16, 160), (1 << 17, 96), (1 << 18, 80), (1 << 19, 40), (1 << 20, 61), (1 << 21, 29), (1 << 22, 17)], ids=str)
def test_fused_two_pass_kernel(ctx, monkeypatch, n, batch):
    data = oracle.get_test_data((n,), numpy.complex64, batch, 4242)
    want = _run_strategy(ctx, monkeypatch, "chain", (n,), batch, data)
    got = _run_strategy(ctx, monkeypatch, "fused", (n,), batch, data)
    if n == 1 << 21:
        # 2048 x 1024: the fused kernel runs the 1024-point pass on the 512-thread tiles (2 x 512 by decimation in time), the
        # chain on the 256-thread ones (radix 16 x 4 x 16): the same transform in another operation order
        assert oracle.difference(want, got, batch) < 5e-7
    else:
        assert numpy.array_equal(want, got), "fused kernel differs from the two-launch chain"
    got_ip = _run_strategy(ctx, monkeypatch, "fused", (n,), batch, data, inplace=True)
    assert numpy.array_equal(got, got_ip), "fused in-place differs"
    # (2^21 has no bit-identity with the chain to lean on: every fourth transform against numpy instead of three)
    for item in (range(0, batch, 4) if n == 1 << 21 else (0, batch // 2, batch - 1)):
        ref = numpy.fft.fft(data[item * n:(item + 1) * n].astype(numpy.complex128))
        g = got[item * n:(item + 1) * n]
        assert numpy.abs(ref - g).sum() / numpy.abs(ref).sum() < 1.1e-6
        assert numpy.abs(ref - g).max() <= 1e-5 * numpy.abs(ref).max()
    back = _run_strategy(ctx, monkeypatch, "fused", (n,), batch, got, inverse=True)
    assert oracle.difference(data, back, batch) < 1.1e-6


@pytest.mark.parametrize("dtype", [numpy.complex64, numpy.float32, numpy.complex128, numpy.float64], ids=str)
def test_fused_2d_1024(ctx, monkeypatch, dtype):
    """BASELINE config 3 shape through the fused kernel's 2-D form (two transposing passes): same thresholds as the chain, in
    place == out of place, inverse round trip, interleaved and split planes."""
    double = numpy.dtype(dtype).itemsize == (8 if numpy.dtype(dtype).kind == "f" else 16)
    shape, batch = (1024, 1024), (29 if double else 57)
    tol, tol_max, tol_chain = (1e-11, 1e-11, 1e-14) if double else (1.1e-6, 1e-5, 5e-7)
    split = numpy.dtype(dtype).kind == "f"
    data = oracle.get_test_data(shape, dtype, batch, 1003)
    bufs_in = data if split else (data,)
    monkeypatch.setenv("PYFFT_AMD_STRATEGY", "fused" if split else "auto")   # (split planes default to the pipelined chunks)
    plan = ctx.getPlan(shape, dtype=dtype)
    assert plan.strategy(batch)[0] == "fused2"
    gin = [ctx.toGpu(x) for x in bufs_in]
    gout = [ctx.allocate(x.shape, x.dtype) for x in bufs_in]
    plan.execute(*gin, *gout, batch=batch)
    for g, x in zip(gin, bufs_in):
        assert numpy.array_equal(g.get(), x)
    out = [g.get() for g in gout]
    plan.execute(*gin, batch=batch)                    # in place
    for g, o in zip(gin, out):
        assert numpy.array_equal(g.get(), o)
    for item in (0, batch // 2, batch - 1):
        sl = slice(item * shape[0], (item + 1) * shape[0])
        x = (bufs_in[0][sl].astype(numpy.float64) + 1j * bufs_in[1][sl]) if split else bufs_in[0][sl].astype(numpy.complex128)
        ref = numpy.fft.fft2(x)
        got = (out[0][sl].astype(numpy.float64) + 1j * out[1][sl]) if split else out[0][sl]
        assert numpy.abs(ref - got).sum() / numpy.abs(ref).sum() < tol
        assert numpy.abs(ref - got).max() <= tol_max * numpy.abs(ref).max()
    plan.execute(*gin, batch=batch, inverse=True)      # back, in place
    for g, x in zip(gin, bufs_in):
        back = g.get()
        assert numpy.abs(back - x).sum() / numpy.abs(x).sum() < tol
    monkeypatch.setenv("PYFFT_AMD_STRATEGY", "chain")
    plan_c = ctx.getPlan(shape, dtype=dtype)
    assert plan_c.strategy(batch)[0] == "chain"
    gin = [ctx.toGpu(x) for x in bufs_in]
    plan_c.execute(*gin, batch=batch)
    for g, o in zip(gin, out):
        c = g.get()
        assert numpy.abs(c - o).sum() / numpy.abs(c).sum() < tol_chain


@pytest.mark.parametrize("side,batch", [(512, 136), (2048, 15)], ids=str)   # (above the 256 MiB-per-side chain threshold)
def test_fused_2d_other_squares(ctx, monkeypatch, side, batch):
    """The 2-D form of the fused kernel for the 512 (256-thread tiles) and 2048 (512-thread tiles) squares, fp32 interleaved."""
    shape = (side, side)
    data = oracle.get_test_data(shape, numpy.complex64, batch, 1006)
    monkeypatch.setenv("PYFFT_AMD_STRATEGY", "auto")
    plan = ctx.getPlan(shape, dtype=numpy.complex64)
    assert plan.strategy(batch)[0] == "fused2"
    a = ctx.toGpu(data)
    b = ctx.allocate(data.shape, data.dtype)
    plan.execute(a, b, batch=batch)
    assert numpy.array_equal(a.get(), data)
    out = b.get()
    plan.execute(a, batch=batch)
    assert numpy.array_equal(a.get(), out)
    for item in (0, batch // 2, batch - 1):
        sl = slice(item * side, (item + 1) * side)
        ref = numpy.fft.fft2(data[sl].astype(numpy.complex128))
        assert numpy.abs(ref - out[sl]).sum() / numpy.abs(ref).sum() < 1.1e-6
        assert numpy.abs(ref - out[sl]).max() <= 1e-5 * numpy.abs(ref).max()
    plan.execute(a, batch=batch, inverse=True)
    assert oracle.difference(data, a.get(), batch) < 1.1e-6


def test_fused_two_pass_kernel_fp64(ctx, monkeypatch):
    """fp64 N = 2^20 (1024 x 1024 on the 512-thread tiles): the fused kernel gives the bits of the two-launch chain."""
    n, batch = 1 << 20, 31
    data = oracle.get_test_data((n,), numpy.complex128, batch, 4243)
    outs = {}
    for strat in ("chain", "fused"):
        monkeypatch.setenv("PYFFT_AMD_STRATEGY", strat)
        plan = ctx.getPlan((n,), dtype=numpy.complex128)
        assert plan.strategy(batch)[0] == {"fused": "fused2", "chain": "chain"}[strat]
        a = ctx.toGpu(data)
        b = ctx.allocate(data.shape, data.dtype)
        plan.execute(a, b, batch=batch)
        assert numpy.array_equal(a.get(), data)
        outs[strat] = b.get()
        plan.execute(a, batch=batch)
        assert numpy.array_equal(a.get(), outs[strat])
    assert numpy.array_equal(outs["chain"], outs["fused"])
    ref = oracle.numpy_fft(numpy.fft.fft, data[:2 * n], 2)
    assert oracle.difference(ref, outs["fused"][:2 * n], 2) < 1e-14
    # split planes
    re, im = oracle.get_test_data((n,), numpy.float64, batch, 4244)
    monkeypatch.setenv("PYFFT_AMD_STRATEGY", "fused")
    plan = ctx.getPlan((n,), dtype=numpy.float64)
    assert plan.strategy(batch)[0] == "fused2"
    a, b = ctx.toGpu(re), ctx.toGpu(im)
    plan.execute(a, b, batch=batch)
    sl = slice((batch - 1) * n, batch * n)
    ref = numpy.fft.fft(re[sl] + 1j * im[sl])
    got = a.get()[sl] + 1j * b.get()[sl]
    assert numpy.abs(ref - got).sum() / numpy.abs(ref).sum() < 1e-14


@pytest.mark.parametrize("shape,batch", [((1 << 17,), 259), ((1 << 16,), 515), ((4096, 8), 1030)], ids=str)
def test_pipelined_chunks(ctx, monkeypatch, shape, batch):
    data = oracle.get_test_data(shape, numpy.complex64, batch, 777)
    want = _run_strategy(ctx, monkeypatch, "chain", shape, batch, data)
    got = _run_strategy(ctx, monkeypatch, "pipelined", shape, batch, data)
    assert numpy.array_equal(want, got), "pipelined execution differs from the plain chain"
    got_ip = _run_strategy(ctx, monkeypatch, "pipelined", shape, batch, data, inplace=True)
    assert numpy.array_equal(want, got_ip)
    ref = oracle.numpy_fft(numpy.fft.fftn, data[:shape[0] * 2], 2)
    assert oracle.difference(ref, got[:shape[0] * 2], 2) < 1.1e-6


@pytest.mark.parametrize("n,batch,strat", [(1 << 16, 520, "pipelined"), (1 << 20, 61, "fused"), (1 << 18, 80, "fused"), (1 << 22, 15, "fused"), (1 << 21, 30, "fused")], ids=str)
def test_split_plane_strategies(ctx, monkeypatch, n, batch, strat):
    """float32 split planes through the chunked / fused strategies (the plan's temp buffer is interleaved even
    though the user buffers are planes): bit-identical to the plain chain, within tolerance of numpy."""
    re, im = oracle.get_test_data((n,), numpy.float32, batch, 99)
    outs = {}
    for s in ("chain", strat):
        monkeypatch.setenv("PYFFT_AMD_STRATEGY", s)
        plan = ctx.getPlan((n,), dtype=numpy.float32)
        assert plan.strategy(batch)[0] == {"fused": "fused2", "chain": "chain", "pipelined": "pipelined"}[s]
        a, b = ctx.toGpu(re), ctx.toGpu(im)
        c, d = ctx.allocate(re.shape, re.dtype), ctx.allocate(im.shape, im.dtype)
        plan.execute(a, b, c, d, batch=batch)
        assert numpy.array_equal(a.get(), re) and numpy.array_equal(b.get(), im)
        outs[s] = (c.get(), d.get())
        plan.execute(a, b, batch=batch)                  # in place
        assert numpy.array_equal(a.get(), outs[s][0]) and numpy.array_equal(b.get(), outs[s][1])
    if n != 1 << 21:   # (2048 x 1024 fused: another operation order than the chain, see test_fused_two_pass_kernel)
        assert numpy.array_equal(outs["chain"][0], outs[strat][0]) and numpy.array_equal(outs["chain"][1], outs[strat][1])
    for item in (0, batch - 1):
        sl = slice(item * n, (item + 1) * n)
        ref = numpy.fft.fft(re[sl].astype(numpy.float64) + 1j * im[sl])
        got = outs[strat][0][sl].astype(numpy.float64) + 1j * outs[strat][1][sl]
        assert numpy.abs(ref - got).sum() / numpy.abs(ref).sum() < 1.1e-6


GOLDEN = os.path.join(ROOT, "tests", "golden", "fft_vectors.npz")


def _golden_index():
    z = numpy.load(GOLDEN)
    return json.loads(bytes(z["index_json"]).decode())


@pytest.mark.parametrize("entry", _golden_index(), ids=lambda e: e["key"])
def test_golden_fixture_through_hip(ctx, entry):
    """Every entry of tests/golden/fft_vectors.npz (seeded input + numpy.fft.fftn of the complex128-upcast input,
    generated by tests/golden/make_golden.py) through Plan()/execute(): forward against the stored output with the
    reference's thresholds (test/test_errors.py:20-23) and the north star's max-norm bound, then inverse against the input."""
    z = numpy.load(GOLDEN)
    data, want = z[entry["key"] + "_in"], z[entry["key"] + "_fw"]
    shape, batch = tuple(entry["shape"]), entry["batch"]
    dtype = numpy.dtype(entry["dtype"])
    eps, mx = (1.1e-6, 1e-5) if dtype == numpy.complex64 else (1e-11, 1e-10)
    plan = ctx.getPlan(shape, dtype=dtype, context=ctx.context)
    a = ctx.toGpu(data)
    b = ctx.allocate(data.shape, data.dtype)
    plan.execute(a, b, batch=batch)
    got = ctx.fromGpu(b, data.shape, data.dtype)
    assert oracle.difference(want, got.astype(numpy.complex128), batch) < eps
    assert numpy.abs(got - want).max() <= mx * numpy.abs(want).max()
    assert numpy.array_equal(ctx.fromGpu(a, data.shape, data.dtype), data)
    plan.execute(b, inverse=True, batch=batch)
    assert oracle.difference(data, ctx.fromGpu(b, data.shape, data.dtype), batch) < eps


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

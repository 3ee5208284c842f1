"""API-contract tests on the GPU: counterpart of the reference's test/test_functionality.py
(TestPlan :7-139, CudaPlan :142-165), one test per reference test, both precisions."""
import numpy
import pytest

pytestmark = pytest.mark.gpu

PRECISIONS = [(numpy.float32, numpy.complex64), (numpy.float64, numpy.complex128)]


@pytest.fixture(params=PRECISIONS, ids=["float", "double"])
def prec(request):
    return request.param


def test_shapes(ctx, prec):                       # test_functionality.py:12-15
    scalar, _ = prec
    for shape in [16, (16,), (16, 16), (16, 16, 16)]:
        ctx.getPlan(shape, dtype=scalar, context=ctx.context)


def test_types(ctx, prec):                        # :17-22
    for dtype in prec:
        ctx.getPlan((16, 16), dtype=dtype, context=ctx.context)


def test_execute_signature_split(ctx, prec):      # :24-36
    dtype = prec[0]
    plan = ctx.getPlan((16,), dtype=dtype, context=ctx.context)
    a, b, c, d = (ctx.toGpu(numpy.ones(16, dtype=dtype)) for _ in range(4))
    plan.execute(a, b)
    plan.execute(a, b, c, d)
    plan.execute(a, b, a, b)


def test_execute_signature_interleaved(ctx, prec):  # :38-51
    dtype = prec[1]
    plan = ctx.getPlan((16,), dtype=dtype, context=ctx.context)
    a = ctx.toGpu(numpy.ones(16, dtype=dtype))
    b = ctx.toGpu(numpy.ones(16, dtype=dtype))
    plan.execute(a)
    plan.execute(a, b)
    plan.execute(a, a)
    with pytest.raises(TypeError):
        plan.execute(a, b, a, b, inverse=True)


def test_normalize(ctx, prec):                    # :53-77
    dtype = prec[1]
    data = numpy.ones(16, dtype=dtype)
    for normalize in [True, False]:
        plan = ctx.getPlan(data.shape, normalize=normalize, dtype=dtype, context=ctx.context)
        a = ctx.toGpu(data)
        plan.execute(a)
        res = ctx.fromGpu(a, data.shape, data.dtype)
        assert numpy.sum(numpy.abs(numpy.fft.fft(data) - res)) / data.size < 1e-6
        plan.execute(a, inverse=True)
        res = ctx.fromGpu(a, data.shape, data.dtype)
        coeff = 1 if normalize else data.size
        assert numpy.sum(numpy.abs(data * coeff - res)) / data.size < 1e-6


def test_scale(ctx, prec):                        # :79-100
    dtype = prec[1]
    data = numpy.ones(16, dtype=dtype)
    for scale in [1.0, 10.0]:
        plan = ctx.getPlan(data.shape, scale=scale, dtype=dtype, context=ctx.context, normalize=True)
        a = ctx.toGpu(data)
        plan.execute(a)
        res = ctx.fromGpu(a, data.shape, data.dtype)
        assert numpy.sum(numpy.abs(numpy.fft.fft(data) * scale - res)) / data.size < 1e-6
        plan.execute(a, inverse=True)
        res = ctx.fromGpu(a, data.shape, data.dtype)
        assert numpy.sum(numpy.abs(data - res)) / data.size < 1e-6


def test_fast_math(ctx, prec):                    # :102-115 (8192 = two-pass path with temp)
    dtype = prec[1]
    data = numpy.ones(8192, dtype=dtype)
    for fast_math in [True, False]:
        plan = ctx.getPlan(data.shape, normalize=True, dtype=dtype, context=ctx.context, fast_math=fast_math)
        a = ctx.toGpu(data)
        plan.execute(a)
        plan.execute(a, inverse=True)
        res = ctx.fromGpu(a, data.shape, data.dtype)
        assert numpy.sum(numpy.abs(data - res)) / data.size < 1e-6


def test_allocation(ctx, prec):                   # :117-121
    plan = ctx.getPlan((32, 32, 32), dtype=prec[1], context=ctx.context)
    a = ctx.toGpu(numpy.ones((32, 32, 32), dtype=prec[1]))
    plan.execute(a)


def test_precreated_context(ctx, prec):           # :123-127
    plan = ctx.getPlan((16,), dtype=prec[1], context=ctx.context)
    a = ctx.toGpu(numpy.ones((16,), dtype=prec[1]))
    plan.execute(a)


def test_wrong_data_size(ctx, prec):              # :129-130
    with pytest.raises(ValueError):
        ctx.getPlan((17,), dtype=prec[1])


def test_wrong_data_type(ctx):                    # :132-133
    with pytest.raises(ValueError):
        ctx.getPlan((16,), dtype=numpy.int32)


def test_wrong_shape(ctx, prec):                  # :135-139
    with pytest.raises(ValueError):
        ctx.getPlan((16, 16, 16, 16), dtype=prec[1])
    with pytest.raises(ValueError):
        ctx.getPlan("16", dtype=prec[1])


def test_mempool(ctx, prec):                      # CudaPlan.testMempool :147-150 (+ it is really used)
    pool = ctx.getMemoryPool()
    n = 1 << 17                                   # two strided passes in both precisions -> needs a temp buffer
    plan = ctx.getPlan((n,), dtype=prec[1], mempool=pool)
    a = ctx.toGpu(numpy.ones(n, dtype=prec[1]))
    plan.execute(a)
    assert pool.calls == 1
    plan.execute(a, inverse=True)
    assert pool.calls == 1            # same batch: temp is reused (plan.py:179-192)
    b = ctx.toGpu(numpy.ones(n * 2, dtype=prec[1]))
    plan.execute(b, batch=2)
    assert pool.calls == 2            # new batch: temp reallocated


def test_external_stream(ctx, prec):              # :152-159
    stream = ctx.hip.Stream()
    plan = ctx.getPlan((32, 32, 32), dtype=prec[1], stream=stream)
    a = ctx.toGpu(numpy.ones((32, 32, 32), dtype=prec[1]))
    ret = plan.execute(a)
    assert ret is stream              # stream given -> asynchronous by default, returns the stream
    stream.synchronize()
    res = ctx.fromGpu(a, (32, 32, 32), prec[1])
    assert abs(res.ravel()[0] - 32 ** 3) < 1e-3 and numpy.abs(res.ravel()[1:]).max() < 1e-3


def test_get_stream(ctx, prec):                   # :161-165
    plan = ctx.getPlan((32, 32, 32), dtype=prec[1])
    a = ctx.toGpu(numpy.ones((32, 32, 32), dtype=prec[1]))
    stream = plan.execute(a, wait_for_finish=False)
    stream.synchronize()
    assert plan.execute(a) is None    # waited -> returns None (plan.py:255-259)


def test_doc_known_answer(ctx):
    """doc/source/index.rst:65-99 and examples/cuda_basic.py: ones((16,16)) -> 256*delta -> ones."""
    data = numpy.ones((16, 16), dtype=numpy.complex64)
    plan = ctx.getPlan((16, 16))
    g = ctx.toGpu(data)
    plan.execute(g)
    res = g.get()
    assert abs(res[0, 0] - 256) < 1e-4
    res[0, 0] = 0
    assert numpy.abs(res).max() < 1e-4
    plan.execute(g, inverse=True)
    assert numpy.abs(g.get() - data).sum() / data.size < 1e-6


def test_torch_tensor_buffers(ctx):
    """Buffers may be torch-ROCm tensors; stream may be a torch stream."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("torch sees no GPU")
    x = torch.randn(4, 1024, dtype=torch.complex64, device="cuda")
    ref = torch.fft.fft(x.to(torch.complex128), dim=1)
    y = torch.empty_like(x)
    s = torch.cuda.current_stream()
    plan = ctx.getPlan((1024,), dtype=numpy.complex64, stream=s)
    plan.execute(x, y, batch=4)
    s.synchronize()
    err = (y.to(torch.complex128) - ref).abs().sum() / ref.abs().sum()
    assert err.item() < 1.1e-6

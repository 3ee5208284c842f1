"""Test harness: counterpart of the reference's test/helpers.py for the HIP backend.

HipContext mirrors helpers.CudaContext (test/helpers.py:29-74): allocate / toGpu / fromGpu /
getPlan / timers / supportsDouble.  Data generation and the error metric come from the oracle
(oracle/pyfft_oracle.py), which only tests may import.
"""
import numpy

COMPLEX_DTYPES = [numpy.complex64, numpy.complex128]
DOUBLE_DTYPES = [numpy.float64, numpy.complex128]


class HipContext(object):

    def __init__(self):
        import pyfft_amd.hip as hip
        self.hip = hip
        if hip.device_count() < 1:
            raise RuntimeError("no HIP device")
        self.context = 0  # device index; what Plan(context=...) accepts

    def allocate(self, shape, dtype):
        return self.hip.DeviceArray(shape, dtype)

    def toGpu(self, data):
        return self.hip.to_gpu(data)

    def fromGpu(self, gpu_buf, target_shape, target_dtype):
        return gpu_buf.get().reshape(target_shape)

    def getMemoryPool(self):
        return CountingPool(self.hip)

    def getPlan(self, *args, **kwds):
        return self.hip.Plan(*args, **kwds)

    def startTimer(self):
        self._start = self.hip.Event().record()
        self._stop = self.hip.Event()

    def stopTimer(self):
        self._stop.record()
        self._stop.synchronize()
        return self._stop.time_since(self._start) / 1000.0

    def supportsDouble(self):
        return True

    def __str__(self):
        return "hip"


class CountingPool(object):
    """mempool stand-in (pycuda.tools.DeviceMemoryPool counterpart): allocate(nbytes)."""

    def __init__(self, hip):
        self.hip = hip
        self.calls = 0

    def allocate(self, nbytes):
        self.calls += 1
        return self.hip.DeviceAllocation(nbytes)


def getDimensions(shape):
    """(test/helpers.py:149-158)"""
    if isinstance(shape, int):
        return shape, 1, 1
    shape = tuple(shape) + (1, 1)
    return shape[0], shape[1], shape[2]


# ---- shared by the GPU test modules (moved here from the per-round files in round 6) ------------------------------------------------
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


EPS_F, MAX_F = 1.1e-6, 1e-5


# ---- persistent executes under stream capture / hipGraph replay -----------------------------------------------------------------
def _tiled_noise(count, dtype, seed):
    rng = numpy.random.default_rng(seed)
    cdt = numpy.dtype(dtype)
    fdt = numpy.float32 if cdt == numpy.complex64 else numpy.float64
    out = numpy.empty(count, cdt)
    for part in ("real", "imag"):
        blk = rng.standard_normal(min(count, (1 << 22) + 17)).astype(fdt)
        setattr(out, part, numpy.resize(blk, count))
    return out


class FakeContext(object):
    """A context without a device: tables are "uploaded" nowhere.  What FFTPlan._select_strategy reads is `machine`."""
    _guard = False

    def __init__(self, machine):
        self.machine = machine
        self.compute_units = machine.compute_units

    def allocate_raw(self, nbytes):
        return 4096

    allocate = allocate_raw

    def upload(self, mem, host):
        pass

    @staticmethod
    def pointer_of(obj):
        return obj

    def capturing(self):
        return False

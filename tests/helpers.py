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

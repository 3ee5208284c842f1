"""ctypes wrapper of oracle/liboracle.so (the plain-C restatement, fft_oracle.c).
TEST INFRASTRUCTURE: importable only from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg."""
import ctypes
import os

import numpy

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")


def available():
    return os.path.exists(_LIB)


def _lib():
    lib = ctypes.CDLL(_LIB)
    lib.oracle_execute.restype = ctypes.c_int
    lib.oracle_execute.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                   ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_double]
    lib.oracle_plan.restype = ctypes.c_int
    lib.oracle_plan.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int,
                                ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]
    return lib


def plan(x, y, z, prec):
    R = (ctypes.c_int * 64)()
    M = (ctypes.c_int64 * 64)()
    S = (ctypes.c_int64 * 64)()
    n = _lib().oracle_plan(x, y, z, prec, R, M, S)
    return [(R[i], M[i], S[i]) for i in range(n)]


def execute(data, xyz, batch=1, inverse=False, divisor=1.0):
    data = numpy.ascontiguousarray(data)
    assert data.dtype in (numpy.complex64, numpy.complex128)
    prec = 1 if data.dtype == numpy.complex128 else 0
    out = numpy.empty_like(data)
    x, y, z = xyz
    rc = _lib().oracle_execute(data.ctypes.data, out.ctypes.data, x, y, z, batch, prec, 1 if inverse else 0, divisor)
    if rc != 0:
        raise ValueError("oracle_execute rejected its arguments")
    return out

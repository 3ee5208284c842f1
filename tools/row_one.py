"""Run one 1-D shape a few times (development tool for rocprofv3 counter passes): row_one.py <log2 n> <c64|c128> [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy
from quick_bench import run
k = int(sys.argv[1]); dt = numpy.complex64 if sys.argv[2] == "c64" else numpy.complex128
run((1 << k,), dt, ((1 << 27) if dt == numpy.complex64 else (1 << 26)) >> k, iters=int(sys.argv[3]) if len(sys.argv) > 3 else 3)

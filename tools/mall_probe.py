"""One-off probes (development tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy
from quick_bench import run
for shape in ((1024, 16), (16, 1024), (512, 32), (256, 64)):
    run(shape, numpy.complex64, (1 << 27) // (shape[0] * shape[1]))

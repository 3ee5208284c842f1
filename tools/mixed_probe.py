"""Throughput of the mixed-radix kernel (csrc/fft_mixed.hip) through Plan(any_size=True): smooth lengths, both precisions, smooth
N-D shapes (one launch per axis).  Development tool; columns as in generic_probe.py."""
import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tools')
import numpy
from generic_probe import run
for n in (1000, 100, 60, 360, 2000, 3000, 1024):
    run((n,), numpy.complex64, (1<<27)//n)
run((1000,), numpy.complex128, 1<<16)
run((100, 100), numpy.complex64, 8192); run((60, 60, 60), numpy.complex64, 512); run((100, 100), numpy.complex128, 4096)
for n in (5000, 10000, 30000, 50000, 196608):
    run((n,), numpy.complex64, (1 << 27) // n)
run((30000,), numpy.complex128, 2048)
for n in (1009, 127, 17, 2039, 513):
    run((n,), numpy.complex64, (1 << 27) // n)
run((1009,), numpy.complex128, 1 << 16); run((1009, 64), numpy.complex64, 2048); run((4099,), numpy.complex64, 1 << 15)
# round 4: smooth N-D shapes in one launch, Bluestein rows up to 5000 points in one launch
for shape, batch in (((60, 60), 1 << 15), ((30, 20, 10), 1 << 14), ((70, 70), 1 << 14), ((12, 20), 1 << 19), ((100, 64), 1 << 14)):
    run(shape, numpy.complex64, batch)
run((70, 70), numpy.complex128, 1 << 13)
for n in (2049, 3001, 4099, 5000):
    run((n,), numpy.complex64, (1 << 27) // n)
run((2500,), numpy.complex128, 1 << 14)

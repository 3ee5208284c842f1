"""Generates the committed golden fixtures.  Run in the build container only (it imports the
reference's pure-python pyfft/kernel_helpers.py from /root/reference; nothing under tests/ reads
/root/reference at test time):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Outputs (data only -- inputs and expected outputs, no reference source text):
  ref_decompositions.json  the reference's own radix decompositions: getGlobalRadixInfo(n) and
                           getRadixArray(n, 0|16) evaluated by the reference code (Python-2 `/`
                           yields floats under Python 3; values are exact for powers of two and
                           are stored as ints)
  fft_vectors.npz          seeded inputs + numpy.fft.fftn outputs on the complex128-upcast input
                           (numpy is the reference's own oracle, test/test_errors.py:5-16,35-36)
"""
import importlib.util
import json
import os
import sys

import numpy

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/pyfft/kernel_helpers.py"

sys.dont_write_bytecode = True
spec = importlib.util.spec_from_file_location("ref_kernel_helpers", REF)
kh = importlib.util.module_from_spec(spec)
spec.loader.exec_module(kh)


def ints(seq):
    return [int(v) for v in seq]


dec = {"getGlobalRadixInfo": {}, "getRadixArray_0": {}, "getRadixArray_16": {}}
for e in range(1, 25):
    n = 2 ** e
    r, r1, r2 = kh.getGlobalRadixInfo(n)
    dec["getGlobalRadixInfo"][str(n)] = [ints(r), ints(r1), ints(r2)]
    if n <= 2048:
        dec["getRadixArray_0"][str(n)] = ints(kh.getRadixArray(n, 0))
    dec["getRadixArray_16"][str(n)] = ints(kh.getRadixArray(n, 16))
with open(os.path.join(HERE, "ref_decompositions.json"), "w") as f:
    json.dump(dec, f, indent=1, sort_keys=True)

cases = []
for n in (8, 16, 64, 256):
    for batch in (1, 3):
        cases.append(((n,), batch))
for n in (1024, 2048, 4096):
    cases.append(((n,), 1))
cases += [((16, 16), 1), ((16, 16), 3), ((128, 32), 1), ((32, 64), 1),
          ((16, 16, 16), 1), ((8, 8, 64), 1), ((32, 16, 8), 1)]

arrays = {}
index = []
for ci, (shape, batch) in enumerate(cases):
    for dt in (numpy.complex64, numpy.complex128):
        seed = 9000 + ci
        rng = numpy.random.default_rng(seed)
        full = (shape[0] * batch,) + tuple(shape[1:])
        fdt = numpy.float32 if dt == numpy.complex64 else numpy.float64
        data = (rng.standard_normal(full).astype(fdt) + 1j * rng.standard_normal(full).astype(fdt)).astype(dt)
        d128 = data.astype(numpy.complex128).reshape((batch,) + tuple(shape))
        fw = numpy.stack([numpy.fft.fftn(d128[i]) for i in range(batch)]).reshape(full)
        key = "c%d_%s_b%d_%s" % (ci, "x".join(map(str, shape)), batch, numpy.dtype(dt).name)
        arrays[key + "_in"] = data
        arrays[key + "_fw"] = fw
        index.append({"key": key, "shape": list(shape), "batch": batch, "dtype": numpy.dtype(dt).name, "seed": seed})
arrays["index_json"] = numpy.frombuffer(json.dumps(index).encode(), dtype=numpy.uint8)
numpy.savez_compressed(os.path.join(HERE, "fft_vectors.npz"), **arrays)
print("wrote", len(index), "vectors;", os.path.getsize(os.path.join(HERE, "fft_vectors.npz")) / 1e6, "MB")

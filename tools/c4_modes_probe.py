#!/usr/bin/env python3
"""Round 6: BASELINE configuration 4 (256^3 complex128 x 64, pipelined pass pairs) measures 13.36 or 13.87 ms per step -- two levels, on one
box and one library (profiles/r06_h_pair_anchor_ab.log).  Does the level follow the placement of the two 16 GiB buffers?  One process:
allocate a pad of k x 2 MiB + 4 KiB x j in front of the buffers, time 6 steps, free everything, next pad.

    python3 tools/c4_modes_probe.py
"""
import os
import sys

import numpy

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfft_amd import _native as N                      # noqa: E402
from pyfft_amd.hip import DeviceArray, Plan, device_props      # noqa: E402


def main():
    p = device_props()
    print(p.name.decode(), p.gcn_arch.decode(), "CUs", p.compute_units, flush=True)
    shape, batch = (256, 256, 256), 64
    n = 256 ** 3 * batch
    plan = Plan(shape, dtype=numpy.complex128)
    for rep in range(2):
        for pad_bytes in (0, 4096, 1 << 16, 1 << 20, (1 << 21) + 4096, 3 << 20, 64 << 20, (1 << 30) + (1 << 20)):
            pad = DeviceArray((max(1, pad_bytes),), numpy.uint8) if pad_bytes else None
            a = DeviceArray((n,), numpy.complex128)
            mid = DeviceArray((max(1, pad_bytes),), numpy.uint8) if pad_bytes else None
            b = DeviceArray((n,), numpy.complex128)
            N.check(N.lib.mifft_memset(a.ptr, 0, 1 << 20, None))
            rng = numpy.random.default_rng(1)
            host = rng.standard_normal(1 << 22).astype(numpy.float64)
            N.check(N.lib.mifft_memcpy_h2d(a.ptr, host.ctypes.data, host.nbytes, None))
            done = host.nbytes
            while done < a.nbytes:
                m = min(done, a.nbytes - done)
                N.check(N.lib.mifft_memcpy_d2d(a.ptr + done, a.ptr, m, None))
                done += m
            plan.timed_execute(2, False, False, batch, [a, None], [b, None])
            ms = [plan.timed_execute(3, False, False, batch, [a, None], [b, None]) / 3 for _ in range(3)]
            print("pad %11d B  in %#x  out %#x   ms per step %s   %s" % (
                pad_bytes, a.ptr, b.ptr, " ".join("%.3f" % m for m in ms), plan.strategy(batch)[0]), flush=True)
            del a, b, pad, mid


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Round 6: how close do the persistent kernels' bounded dependency waits come to their time-out?

A dependency wait polls a counter, sleeps ~1 us (s_sleep 32) and gives up after 2^22 polls -- a few seconds -- by setting the plan's error
word ("results invalid") instead of hanging the GPU (csrc/fft_fused2.hpp, fused_wait_ge).  On a device shared between processes a
work-group can be descheduled for a long time; this tool measures the margin.  `make DEV=1` builds of the library record, per counter
set, the LONGEST wait of any work-group in polls (word 2 of the ticket line); run with PYFFT_AMD_DEV_BUILD=1:

    PYFFT_AMD_DEV_BUILD=1 python3 tools/spin_margin.py            # one process alone, then 2 / 4 / 8 processes sharing the GPU

Every process runs the persistent kernel of BASELINE configuration 2's shape (2^20 points, batch 64 x 14 executes), configuration 5's
(2^22, batch 16) and the two-pair cube kernel (128^3, batch 64) and prints the longest wait it saw.
"""
import ctypes
import os
import subprocess
import sys
import time

import numpy

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CASES = [((1 << 20,), 64, 14), ((1 << 22,), 16, 8), ((128, 128, 128), 64, 14)]
LIMIT = 1 << 22


def worker(tag):
    from pyfft_amd import _native as N
    from pyfft_amd.hip import DeviceArray, Plan
    out = []
    for shape, batch, reps in CASES:
        n = int(numpy.prod(shape)) * batch
        a = DeviceArray((n,), numpy.complex64)
        b = DeviceArray((n,), numpy.complex64)
        N.check(N.lib.mifft_memset(a.ptr, 0, a.nbytes, None))
        plan = Plan(shape if len(shape) > 1 else shape[0], dtype=numpy.complex64, wait_for_finish=False)
        os.environ["PYFFT_AMD_STRATEGY"] = "fused"
        try:
            strat = plan.strategy(batch)
            t0 = time.perf_counter()
            for _ in range(reps):
                plan.execute(a, b, batch=batch)
            plan.finish()
            dt = time.perf_counter() - t0
        finally:
            os.environ.pop("PYFFT_AMD_STRATEGY", None)
        assert strat[0] in ("fused2", "fusedp"), strat
        # word 2 of the ticket line of the counter sets the eager launches alternate between
        base = plan._context.pointer_of(plan._counters)
        worst = 0
        for k in range(2):               # (the two alternating sets; the third one belongs to captured launches)
            w = ctypes.c_uint32(0)
            N.check(N.lib.mifft_memcpy_d2h(ctypes.byref(w), base + k * plan._counter_bytes + 8, 4, None), "d2h")
            worst = max(worst, w.value)
        out.append("%s x %d: longest wait %d polls (%.5f of the time-out), %.1f ms per execute" % (
            "x".join(map(str, shape)), batch, worst, worst / float(LIMIT), dt * 1e3 / reps))
    print("[%s] " % tag + "; ".join(out), flush=True)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "worker":
        return worker(sys.argv[2])
    from pyfft_amd import _native as N
    if N.lib.mifft_has_feature(N.FEATURE_FUSED2X) != 1:
        raise SystemExit("needs the development library: make -C pyfft_amd/csrc DEV=1 and PYFFT_AMD_DEV_BUILD=1")
    print("longest dependency wait of the persistent kernels in polls of ~1 us; the time-out is %d polls" % LIMIT, flush=True)
    for procs in (1, 2, 4, 8):
        print("## %d process(es) on one GPU" % procs, flush=True)
        ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "worker", "%d/%d" % (i, procs)]) for i in range(procs)]
        rc = [p.wait() for p in ps]
        if any(rc):
            raise SystemExit("a worker failed: %r" % (rc,))


if __name__ == "__main__":
    main()

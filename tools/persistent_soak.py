"""Soak of the persistent strategies (fused2 / fused2x / fusedp): every shape class that has such a kernel, both layouts, batches
just beyond the chain threshold, odd batches, batches that fill the ring exactly twice -- each executed three times back to back
(alternating counter sets), out of place and in place, forward and inverse, and compared WHOLE-ARRAY with the chain strategy
(one launch per pass over the whole batch) on the same numbers.  Development tool behind profiles/r04_persistent_soak.log.

    python3 tools/persistent_soak.py [SEED [MAX_CASES]]
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
from pyfft_amd.hip import Plan, DeviceArray
from pyfft_amd import _native as N

c64, c128, f32, f64 = numpy.complex64, numpy.complex128, numpy.float32, numpy.float64
SHAPES = [((1 << k,), dt) for k in range(16, 23) for dt in (c64, f32)] + [((1 << k,), c128) for k in range(16, 22)] + [((1 << k,), f64) for k in range(16, 21)] + \
    [((a, b), dt) for a in (256, 512, 1024, 2048) for b in (256, 512, 1024, 2048) for dt in (c64, c128)] + \
    [((a, b), f32) for a in (256, 512, 1024) for b in (256, 512, 1024)] + [((1024, 1024), f64)] + \
    [((a, b, c), dt) for a in (64, 128) for b in (64, 128) for c in (64, 128) for dt in (c64, c128, f32, f64)] + \
    [(sh, dt) for sh in ((32, 32, 128), (32, 64, 128), (32, 128, 128), (64, 32, 128), (128, 32, 128), (32, 64, 64), (32, 128, 64)) for dt in (c64, c128)]   # round 6


def fill(buf, blk):
    hb = numpy.ascontiguousarray(blk).view(numpy.uint8).reshape(-1)
    n = min(hb.nbytes, buf.nbytes)
    N.check(N.lib.mifft_memcpy_h2d(buf.ptr, hb.ctypes.data, n, None))
    done = n
    while done < buf.nbytes:
        m = min(done, buf.nbytes - done)
        N.check(N.lib.mifft_memcpy_d2d(buf.ptr + done, buf.ptr, m, None))
        done += m
    N.check(N.lib.mifft_device_sync())


def get(bufs):
    out = [numpy.empty(b.shape, b.dtype) for b in bufs]
    for o, b in zip(out, bufs):
        N.check(N.lib.mifft_memcpy_d2h(o.ctypes.data, b.ptr, b.nbytes, None))
    return out[0] if len(out) == 1 else out[0] + 1j * out[1]


def one(shape, dtype, batch, rng):
    dt = numpy.dtype(dtype)
    split = dt.kind == "f"
    size = int(numpy.prod(shape))
    eps = 1e-11 if dt.itemsize * (2 if split else 1) == 16 else 1.1e-6
    nblk = min(batch, 5)                           # (5 distinct transforms, tiled: the tiling period is no divisor of the ring sizes)
    blk = rng.standard_normal((nblk, size)) + 1j * rng.standard_normal((nblk, size))
    n = size * batch
    if split:
        ins = [DeviceArray((n,), dt), DeviceArray((n,), dt)]
        fill(ins[0], blk.real.astype(dt)); fill(ins[1], blk.imag.astype(dt))
    else:
        ins = [DeviceArray((n,), dt)]
        fill(ins[0], blk.astype(dt))
    outs = [DeviceArray((n,), dt) for _ in ins]
    os.environ["PYFFT_AMD_STRATEGY"] = "auto"
    plan = Plan(shape if len(shape) > 1 else shape[0], dtype=dt, wait_for_finish=True)
    strat = plan.strategy(batch)
    if strat[0] not in ("fused2", "fused2x", "fusedp"):
        return strat, None
    os.environ["PYFFT_AMD_STRATEGY"] = "chain"
    cplan = Plan(shape if len(shape) > 1 else shape[0], dtype=dt, wait_for_finish=True)
    assert cplan.strategy(batch)[0] == "chain"
    cplan.execute(*(ins + outs), batch=batch)
    want = get(outs)
    worst = 0.0
    os.environ["PYFFT_AMD_STRATEGY"] = "auto"      # (a plan settles its strategy at its first execute of a batch size)
    for rep in range(3):                           # back to back: the two counter sets alternate
        for b in outs:
            N.check(N.lib.mifft_memset(b.ptr, 0, b.nbytes, None))
        plan.execute(*(ins + outs), batch=batch)
        got = get(outs)
        worst = max(worst, float(numpy.abs(got - want).sum() / numpy.abs(want).sum()))
    # in place, then the inverse in place: back to the input
    x = get(ins)
    plan.execute(*ins, batch=batch)
    worst = max(worst, float(numpy.abs(get(ins) - want).sum() / numpy.abs(want).sum()))
    plan.execute(*ins, batch=batch, inverse=True)
    back = float(numpy.abs(get(ins) - x).sum() / numpy.abs(x).sum())
    plan.finish()
    assert plan._strategy[0] == strat[0], (plan._strategy, strat)
    ok = worst < eps and back < 2 * eps
    if not ok:
        print("  MISMATCH %s %s batch %d %s: vs chain %.2e, round trip %.2e" % (shape, dt.name, batch, strat, worst, back), flush=True)
    plan.close(); cplan.close()
    return strat, ok


if __name__ == "__main__":
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    limit = int(sys.argv[2]) if len(sys.argv) > 2 else 10 ** 6
    rng = numpy.random.default_rng(seed)
    order = rng.permutation(len(SHAPES))
    ran = bad = skipped = 0
    for idx in order:
        shape, dtype = SHAPES[idx]
        dt = numpy.dtype(dtype)
        item = int(numpy.prod(shape)) * dt.itemsize * (2 if dt.kind == "f" else 1)
        b0 = (256 << 20) // item + 1               # just beyond the chain threshold
        for batch in (b0, b0 + int(rng.integers(1, 40)), 2 * b0 + int(rng.integers(0, 9))):
            if ran >= limit:
                break
            strat, ok = one(shape, dtype, batch, rng)
            if ok is None:
                skipped += 1
                print("%-16s %-10s batch %-6d %-34s (no persistent form at this batch)" % ("x".join(map(str, shape)), dt.name, batch, strat[:4]), flush=True)
                continue
            ran += 1
            bad += 0 if ok else 1
            print("%-16s %-10s batch %-6d %-34s %s" % ("x".join(map(str, shape)), dt.name, batch, strat[:4], "ok" if ok else "MISMATCH"), flush=True)
    print("persistent soak: %d cases, %d mismatches, %d without a persistent form (seed %d)" % (ran, bad, skipped, seed))
    sys.exit(1 if bad else 0)

"""Lag / ring / work-list sweep of the persistent two-pass kernels at fixed buffer sizes (development tool behind
profiles/r04_fused_sweep.log): every variant builds a fresh plan in this process with the development switches of
pyfft_amd/_debug.py set, checks sampled transforms against numpy and times back-to-back executes between two HIP events.

    python3 tools/fused_sweep.py [SHAPE DTYPE GIB VARIANTS]...      e.g.  524288 complex64 1 auto,f:14:28,x:4:8:0
    variants: auto | chain | pipelined | f:LAG:RING (fused2 / fusedp) | x:LAG:RING:WT (fusedx, per XCD) | seq | any of them + @ENV=VALUE
    GIB may be a fraction (0.03125 = the reference's 32 MiB protocol)
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
from pyfft_amd.hip import Plan, DeviceArray, Event
from pyfft_amd import _native as N

KEYS = ("PYFFT_AMD_STRATEGY", "PYFFT_AMD_FUSED_RING", "PYFFT_AMD_FUSEDX", "PYFFT_AMD_FUSED3", "PYFFT_AMD_FUSED_WGS", "PYFFT_AMD_PIPE_MB",
        "PYFFT_AMD_FUSED_MEMSET", "PYFFT_AMD_NO_FUSEDX", "PYFFT_AMD_SMALL_FUSED", "MIFFT_PAIR", "MIFFT_NARROW_TILES", "PYFFT_AMD_SPLIT_FUSEDX", "PYFFT_AMD_NO_SPLIT_ROWFIRST", "MIFFT_STORE")


def variant_env(v):
    env = {}
    parts = v.split("@")
    head = parts[0]
    for extra in parts[1:]:
        k, val = extra.split("=")
        env[k] = val
    t = head.split(":")
    if t[0] == "auto":
        pass
    elif t[0] in ("chain", "pipelined"):
        env["PYFFT_AMD_STRATEGY"] = t[0]
    elif t[0] == "f":
        env["PYFFT_AMD_STRATEGY"] = "fused"
        env["PYFFT_AMD_FUSED_RING"] = "%s,%s" % (t[1], t[2])
        env["PYFFT_AMD_FUSED3"] = "%s,%s" % (t[1], t[2])
    elif t[0] == "seq":          # tiny batches: the sequential single-launch work list
        env["PYFFT_AMD_SMALL_FUSED"] = "1"
    elif t[0] == "x":
        env["PYFFT_AMD_STRATEGY"] = "fusedx"
        env["PYFFT_AMD_FUSEDX"] = "%s,%s" % (t[1], t[2])
    else:
        raise ValueError(v)
    return env


def fill(buf, blk):
    hb = blk.view(numpy.uint8).reshape(-1)
    n = min(hb.nbytes, buf.nbytes)
    N.check(N.lib.mifft_memcpy_h2d(buf.ptr, hb.ctypes.data, n, None))
    done = n
    while done < buf.nbytes:
        m = min(done, buf.nbytes - done)
        N.check(N.lib.mifft_memcpy_d2d(buf.ptr + done, buf.ptr, m, None))
        done += m
    N.check(N.lib.mifft_device_sync())


def sweep(shape, dtype, gib, variants, reps=5, iters=10):
    dt = numpy.dtype(dtype)
    split = dt.kind == "f"               # float32 / float64: two scalar planes per side (GIB counts both)
    cbytes = dt.itemsize * (2 if split else 1)
    size = int(numpy.prod(shape))
    batch = max(1, int(gib * (1 << 30)) // (size * cbytes))
    rng = numpy.random.default_rng(7)
    nblk = min(batch, 8)
    cblk = rng.standard_normal((nblk, size)) + 1j * rng.standard_normal((nblk, size))
    if split:
        ins = [DeviceArray((size * batch,), dt), DeviceArray((size * batch,), dt)]
        outs = [DeviceArray((size * batch,), dt), DeviceArray((size * batch,), dt)]
        fill(ins[0], cblk.real.astype(dt)); fill(ins[1], cblk.imag.astype(dt))
        blk = (cblk.real.astype(dt) + 1j * cblk.imag.astype(dt))
    else:
        blk = cblk.astype(dt)
        ins, outs = [DeviceArray((size * batch,), dt)], [DeviceArray((size * batch,), dt)]
        fill(ins[0], blk)
    bufs = ins + outs
    refs = {}
    parts = [numpy.empty(size, dt) for _ in outs]
    for v in variants:
        for k in KEYS:
            os.environ.pop(k, None)
        os.environ.update(variant_env(v))
        N.lib.mifft_debug_set(N.DEBUG_PAIR, int(os.environ.get("MIFFT_PAIR", "0")))     # (library switches: read at import otherwise)
        N.lib.mifft_debug_set(N.DEBUG_NARROW_TILES, int(os.environ.get("MIFFT_NARROW_TILES", "0")))
        N.lib.mifft_debug_set(N.DEBUG_STORE, int(os.environ.get("MIFFT_STORE", "0")))
        try:
            plan = Plan(shape if len(shape) > 1 else shape[0], dtype=dt, wait_for_finish=True)
            for b in outs:
                N.check(N.lib.mifft_memset(b.ptr, 0, b.nbytes, None))
            plan.execute(*bufs, batch=batch)
            worst = 0.0
            for item in sorted(set((0, 1, nblk - 1, batch // 2, batch - 2, batch - 1)) & set(range(batch))):
                for part, b in zip(parts, outs):
                    N.check(N.lib.mifft_memcpy_d2h(part.ctypes.data, b.ptr + item * size * dt.itemsize, size * dt.itemsize, None))
                out = parts[0] + 1j * parts[1] if split else parts[0]
                j = item % nblk
                if j not in refs:
                    refs[j] = numpy.fft.fftn(blk[j].reshape(shape).astype(numpy.complex128)).reshape(-1)
                worst = max(worst, float(numpy.abs(out - refs[j]).sum() / numpy.abs(refs[j]).sum()))
            st = plan._context.getQueue()
            best = 1e9
            for _ in range(reps):
                e0 = Event().record(st)
                for _ in range(iters):
                    plan.execute(*bufs, batch=batch, wait_for_finish=False)
                e1 = Event().record(st)
                e1.synchronize()
                best = min(best, e1.time_since(e0) / iters)
            plan.finish()
            frac = 2.0 * size * batch * cbytes / (best * 1e-3) / 8e12
            print("%-16s %-10s x %-6d %-28s %-40s %9.3f ms  %.3f  err %.1e" % (
                "x".join(str(s) for s in shape), dt.name, batch, v, str(plan.strategy(batch)[:5]), best, frac, worst), flush=True)
            plan.close()
        except Exception as e:      # a variant the shape has no kernel for: say so and go on
            print("%-16s %-10s x %-6d %-28s FAILED %r" % ("x".join(str(s) for s in shape), dt.name, batch, v, e), flush=True)
    for k in KEYS:
        os.environ.pop(k, None)


if __name__ == "__main__":
    args = sys.argv[1:]
    while len(args) >= 4:
        shape = tuple(int(t) for t in args[0].split("x"))
        sweep(shape, args[1], float(args[2]), args[3].split(","))
        args = args[4:]

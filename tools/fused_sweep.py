"""Lag / ring / work-list sweep of the persistent two-pass kernels at fixed buffer sizes (development tool behind
profiles/r04_fused_sweep.log): every variant builds a fresh plan in this process with the development switches of
pyfft_amd/_debug.py set, checks sampled transforms against numpy and times back-to-back executes between two HIP events.

    python3 tools/fused_sweep.py [SHAPE DTYPE GIB VARIANTS]...      e.g.  524288 complex64 1 auto,f:14:28,x:4:8:0
    variants: auto | chain | pipelined | fused | f:LAG:RING (fused2 / fusedp) | x:LAG:RING:WT (fusedx, per XCD) | seq | any of them + @ENV=VALUE
    GIB may be a fraction (0.03125 = the reference's 32 MiB protocol)

    python3 tools/fused_sweep.py --emit OUT.json [--gib 2]
        regenerates the planner's tuning table (pyfft_amd/tuning_gfx950.json, round 5): for every rule and every shape class the rule
        answers for, the rule's persistent launch (PYFFT_AMD_STRATEGY=fused: the table's own tile width, work-groups per CU and ring
        rule) against the pipelined chunks at GIB per side; writes the table with the fractions under `measured` and with `on_request`
        set where the persistent form loses on this device (cleared where it wins), everything else as loaded.
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
from pyfft_amd.hip import Plan, DeviceArray, Event
from pyfft_amd import _native as N

KEYS = ("PYFFT_AMD_STRATEGY", "PYFFT_AMD_FUSED_RING", "PYFFT_AMD_FUSEDX", "PYFFT_AMD_FUSED3", "PYFFT_AMD_FUSED_WGS", "PYFFT_AMD_PIPE_MB", "PYFFT_AMD_PIPE_STREAMS",
        "PYFFT_AMD_FUSED_MEMSET", "PYFFT_AMD_NO_FUSEDX", "PYFFT_AMD_SMALL_FUSED", "MIFFT_PAIR", "MIFFT_NARROW_TILES", "PYFFT_AMD_SPLIT_FUSEDX", "PYFFT_AMD_NO_SPLIT_ROWFIRST", "MIFFT_STORE", "PYFFT_AMD_NO_OOP_ND", "MIFFT_NO_ND2", "PYFFT_AMD_NO_ND_GENERIC", "MIFFT_NO_WAVE", "MIFFT_FORCE_WAVE", "MIFFT_ALT_ROWS")


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
    elif t[0] in ("chain", "pipelined", "fused"):
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


def sweep(shape, dtype, gib, variants, reps=5, iters=10, quiet=False):
    """[(variant, strategy tuple or None, ms per execute, fraction of 8 TB/s, worst L1-relative error vs numpy)]"""
    results = []
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
        N.lib.mifft_debug_set(N.DEBUG_NO_ND2, int(os.environ.get("MIFFT_NO_ND2", "0")))
        N.lib.mifft_debug_set(N.DEBUG_NO_WAVE, int(os.environ.get("MIFFT_NO_WAVE", "0")))
        N.lib.mifft_debug_set(N.DEBUG_FORCE_WAVE, int(os.environ.get("MIFFT_FORCE_WAVE", "0")))
        N.lib.mifft_debug_set(N.DEBUG_ALT_ROWS, int(os.environ.get("MIFFT_ALT_ROWS", "0")))
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
            results.append((v, tuple(plan.strategy(batch)[:5]), best, frac, worst))
            print("%-16s %-10s x %-6d %-28s %-40s %9.3f ms  %.3f  err %.1e" % (
                "x".join(str(s) for s in shape), dt.name, batch, v, str(plan.strategy(batch)[:5]), best, frac, worst), flush=True)
            plan.close()
        except Exception as e:      # a variant the shape has no kernel for: say so and go on
            results.append((v, None, None, None, None))
            print("%-16s %-10s x %-6d %-28s FAILED %r" % ("x".join(str(s) for s in shape), dt.name, batch, v, e), flush=True)
    for k in KEYS:
        os.environ.pop(k, None)
    return results


def rule_cases(tuning, rule):
    """(shape, dtype name) of every shape class `rule` answers for under the default switches (first match wins: a later rule only
    gets what the earlier ones leave)."""
    out = []
    precs = [rule["precision"]] if rule.get("precision") else ["f32", "f64"]
    lays = [rule["layout"]] if rule.get("layout") else ["interleaved", "split"]
    when = rule.get("when", {})
    narrow, rowfirst = bool(when.get("narrow_tiles", False)), bool(when.get("rowfirst", True))
    if narrow or not rowfirst:
        return out            # (forms behind development switches: not re-measured here)
    for prec in precs:
        for lay in lays:
            dt = {("f32", "interleaved"): "complex64", ("f32", "split"): "float32", ("f64", "interleaved"): "complex128",
                  ("f64", "split"): "float64"}[(prec, lay)]
            if rule["kind"] == "1d":
                for L0 in rule["L0"]:
                    for L1 in rule["L1"]:
                        cls = {"kind": "1d", "precision": prec, "layout": lay, "L0": L0, "L1": L1, "M": L1}
                        if L0 >= L1 and tuning.match(cls, narrow, rowfirst) is rule:
                            out.append(((L0 * L1,), dt))
            elif rule["kind"] == "2d":
                for ny in rule["ny"]:
                    for nx in rule["nx"]:
                        cls = {"kind": "2d", "precision": prec, "layout": lay, "ny": ny, "nx": nx}
                        if tuning.match(cls, narrow, rowfirst) is rule:
                            out.append(((ny, nx), dt))
            elif rule["kind"] == "3d":
                for shape in ((128, 128, 128), (64, 64, 64), (64, 128, 128)):
                    out.append((shape, dt))
    return out


def write_table(path, table):
    """The tuning table in the layout of pyfft_amd/tuning_gfx950.json: one key per line, one rule per line."""
    import json
    out = "{\n"
    for k, v in table.items():
        if k != "rules":
            out += " %s: %s,\n" % (json.dumps(k), json.dumps(v))
    out += ' "rules": [\n' + ",\n".join("  " + json.dumps(r) for r in table["rules"]) + "\n ]\n}\n"
    with open(path, "w") as f:
        f.write(out)


def emit(path, gib):
    """Re-measure every rule of the tuning table against the pipelined chunks and write the table back (see the module docstring)."""
    import copy
    import json
    from pyfft_amd import tuning as T
    base = T.default()
    table = copy.deepcopy(base.table)
    for rule, new in zip(base.rules, table["rules"]):
        measured = []
        for shape, dt in rule_cases(base, rule):
            res = sweep(shape, dt, gib, ["fused", "pipelined"], reps=3, iters=8)
            by = {v: (st, frac) for v, st, ms, frac, err in res}
            fused, pipe = by.get("fused", (None, None)), by.get("pipelined", (None, None))
            if fused[0] is None or fused[0][0] not in ("fused2", "fusedp") or pipe[1] is None:
                continue          # (no persistent launch came out: batch too small for the ring at this size, or no kernel)
            measured.append({"shape": list(shape), "dtype": dt, "gib_per_side": gib, "persistent": round(fused[1], 4),
                             "pipelined": round(pipe[1], 4), "lag_ring_grid": list(fused[0][1:4])})
        if measured:
            new["measured"] = measured
            # (hysteresis: a default only changes on a clear verdict -- every shape of the rule 2 % ahead, or every shape 2 % behind)
            if all(m["persistent"] >= 1.02 * m["pipelined"] for m in measured):
                new.pop("on_request", None)
            elif all(m["persistent"] <= 0.98 * m["pipelined"] for m in measured):
                new["on_request"] = True
        print("rule %-90s %s" % (rule["name"][:90], "on request" if new.get("on_request") else "default"), flush=True)
    write_table(path, table)
    changed = [r["name"] for r, n in zip(base.rules, table["rules"]) if bool(r.get("on_request")) != bool(n.get("on_request"))]
    print("wrote %s; rules whose default changed: %s" % (path, changed or "none"))


if __name__ == "__main__":
    args = sys.argv[1:]
    if args and args[0] == "--emit":
        gib = float(args[args.index("--gib") + 1]) if "--gib" in args else 2.0
        emit(args[1], gib)
        args = []
    while len(args) >= 4:
        shape = tuple(int(t) for t in args[0].split("x"))
        sweep(shape, args[1], float(args[2]), args[3].split(","))
        args = args[4:]

"""Round 5: is the default kernel of every single-launch plan the fastest one the library has?  1-D lengths 2 ... 32768 and the N-D shapes of
the fixed-shape tables, in all four dtypes (interleaved and split-complex), at two buffer sizes: the default plan against the library's
alternatives (MIFFT_NO_WAVE / MIFFT_FORCE_WAVE for the short rows, MIFFT_NO_ND2 for the N-D shapes in the split layouts).  A row is flagged
when an alternative is more than 5 % ahead.  Development tool behind profiles/r05_single_pass_alternatives.log.

    python3 tools/single_pass_sweep.py rows|nd-split [GIB ...]      (default sizes: 1 and 0.03125 GiB per side)
"""
import contextlib
import io
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fused_sweep
import gen_nd2_tables


def measure(shape, dt, gib, variants):
    with contextlib.redirect_stdout(io.StringIO()):
        res = fused_sweep.sweep(shape, dt, gib, variants, reps=3, iters=10)
    return [r[3] for r in res]


def report(label, shape, dt, gibs, variants):
    row = "%-10s %-14s" % (dt, "x".join(str(n) for n in shape))
    flag = ""
    for gib in gibs:
        fr = measure(shape, dt, gib, variants)
        row += "   %6.3f GiB " % gib + " / ".join("%.3f" % f if f is not None else " n/a " for f in fr)
        alts = [f for f in fr[1:] if f is not None]
        if fr[0] is not None and alts and max(alts) > 1.05 * fr[0]:
            flag = "   <-- an alternative is ahead"
    print(row + flag, flush=True)


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "rows"
    gibs = [float(a) for a in sys.argv[2:]] or [1.0, 0.03125]
    if what == "rows":
        print("# default / MIFFT_NO_WAVE=1 / MIFFT_FORCE_WAVE=1")
        for dt in ("complex64", "complex128", "float32", "float64"):
            for k in range(1, 16):
                report("rows", (1 << k,), dt, gibs, ["auto", "auto@MIFFT_NO_WAVE=1", "auto@MIFFT_FORCE_WAVE=1"])
    elif what == "nd-split":
        print("# default / MIFFT_NO_ND2=1")
        for prec, dt in (("f32", "float32"), ("f64", "float64")):
            for (x, y, z) in gen_nd2_tables.shapes(prec):
                shape = tuple(n for n in (z, y, x) if n > 1) if z > 1 else ((y, x) if y > 1 else (x,))
                if len(shape) >= 2:
                    report("nd", shape, dt, gibs, ["auto", "auto@MIFFT_NO_ND2=1"])


if __name__ == "__main__":
    main()

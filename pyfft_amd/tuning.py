"""The planner's measured constants as data (round 5): pyfft_amd/tuning_gfx950.json.

`FFTPlan._select_strategy` used to be 110 lines of nested special cases keyed on literals; every one of them was a measurement
(a log under profiles/), none could be regenerated.  They now live in one table: which shape classes run a persistent launch
(`rules`, first match wins), on which first-pass tile width and with how many work-groups per CU, from which transform size on;
the fractions of the last-level cache that rings, chunks and thresholds are (`cache_fractions`); the lag / ring rule (`ring_rule`).
The planner matches a plan's shape class against the table and scales by the device's own cache size and CU count
(pyfft_amd/machine.py).  `tools/fused_sweep.py --emit` re-measures the rules on a GPU box and rewrites the table.

The reference has no counterpart: its only device-dependent choices are block and grid limits (pyfft/cuda.py:72-83,
pyfft/kernel.py:46-83).
"""
import json
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_PATH = os.path.join(_HERE, "tuning_gfx950.json")


class Tuning(object):
    def __init__(self, table, source=None):
        self.table = table
        self.source = source
        self.fractions = {k: (int(v[0]), int(v[1])) for k, v in table["cache_fractions"].items() if isinstance(v, list)}
        self.ring_rule = table["ring_rule"]
        self.pipelined = table["pipelined"]
        self.development = table["development"]
        self.rules = list(table["rules"])
        # one-launch N-D shapes that run the run-time-shaped kernel: {precision: (always, big)} as sets of (x, y, z)
        ng = table.get("nd_generic", {})
        self.nd_generic = {prec: (frozenset(tuple(t) for t in ng.get(prec, {}).get("always", [])),
                                  frozenset(tuple(t) for t in ng.get(prec, {}).get("big", []))) for prec in ("f32", "f64", "f32_split", "f64_split")}
        for r in self.rules:
            for key in ("name", "kind", "strategy", "extent0", "cols0", "per_cu"):
                if key not in r:
                    raise ValueError("tuning table %s: rule %r lacks %r" % (source, r.get("name"), key))

    @classmethod
    def load(cls, path=None):
        path = path or os.environ.get("PYFFT_AMD_TUNING") or DEFAULT_PATH
        with open(path) as f:
            return cls(json.load(f), path)

    def fraction(self, name, value):
        num, den = self.fractions[name]
        return int(value) * num // den

    def nd_runs_generic(self, f64, xyz, big_launch, split=False):
        """The plan's N-D pass of shape (x, y, z) takes the run-time-shaped kernel (pass variant 1)."""
        always, big = self.nd_generic[("f64" if f64 else "f32") + ("_split" if split else "")]
        return xyz in always or (big_launch and xyz in big)

    def match(self, shape_class, narrow_tiles, rowfirst):
        """First rule whose conditions hold for the plan's shape class (a dict: kind, precision, layout and the class's lengths --
        L0 / L1 of a two-pass 1-D plan, ny / nx of a ROW + COL 2-D plan), or None."""
        if shape_class is None:
            return None
        for r in self.rules:
            if r["kind"] != shape_class["kind"]:
                continue
            if r.get("precision") not in (None, shape_class["precision"]) or r.get("layout") not in (None, shape_class["layout"]):
                continue
            if any(k in r and shape_class.get(k) not in r[k] for k in ("L0", "L1", "ny", "nx")):
                continue
            if r.get("order") == "L0>=L1" and not shape_class["L0"] >= shape_class["L1"]:
                continue
            if r.get("square") and shape_class.get("ny") != shape_class.get("nx"):
                continue
            if "has_side" in r and r["has_side"] not in (shape_class.get("ny"), shape_class.get("nx")):
                continue
            when = r.get("when", {})
            if "narrow_tiles" in when and bool(when["narrow_tiles"]) != bool(narrow_tiles):
                continue
            if "rowfirst" in when and bool(when["rowfirst"]) != bool(rowfirst):
                continue
            return r
        return None


_default = None


def default():
    """The table every plan uses (loaded once; PYFFT_AMD_TUNING names another file)."""
    global _default
    if _default is None:
        _default = Tuning.load()
    return _default

"""Generator of tests/golden/strategy_snapshot.json.gz: what FFTPlan._select_strategy answers for a grid of (machine, shape, dtype, batch)
-- every shape of profiles/r04_long_1d_sizes.log / r04_second_batch_shapes.log / r04_t_tail_survey.log at several buffer sizes, on the
full part, a one-XCD partition and a device without a last-level cache.  Written with the planner of commit 2d3a19f (end of round 4),
BEFORE its thresholds moved into pyfft_amd/tuning_gfx950.json: tests/test_host.py::test_strategy_snapshot keeps the table-driven
planner on the same answers.  Written against a `make DEV=1` build of the library (the per-XCD work lists appear under
MIFFT_NARROW_TILES=1; the test skips those rows on the default build).  Runs without a GPU (the library's support queries need none):

    python tests/golden/make_strategy_snapshot.py            # rewrites the json
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import numpy

MACHINES = {"full": (256, 8, 4 << 20, 256 << 20), "partition": (32, 1, 4 << 20, 32 << 20), "no_llc": (64, 2, 4 << 20, 0)}
DTYPES = ("complex64", "complex128", "float32", "float64")
SIDES_MIB = (32, 256, 257, 512, 1024, 2048, 8192)     # buffer size per side the batch is derived from


def shapes():
    out = [(1 << k,) for k in range(10, 25)]
    sides = (128, 256, 512, 1024, 2048, 4096)
    out += [(a, b) for a in sides for b in sides]
    out += [(64, 16384), (16384, 64), (32, 1024), (8192, 8192)]
    cube = (64, 128, 256)
    out += [(a, b, c) for a in cube for b in cube for c in cube]
    out += [(16, 16, 128), (32, 32, 128), (32, 32, 2048), (2048, 32, 32), (512, 512, 512), (16, 16, 16), (32, 32, 32), (8, 64, 64), (128, 128, 1024)]
    return out


def cases():
    for shape in shapes():
        size = int(numpy.prod(shape))
        for dt in DTYPES:
            csize = 8 if dt in ("complex64", "float32") else 16
            seen = set()
            for mib in SIDES_MIB:
                batch = max(1, (mib << 20) // (size * csize))
                for b in (batch, batch - 1 if batch > 2 else batch):
                    if b not in seen:
                        seen.add(b)
                        yield shape, dt, b


# development switches the planner's answers depend on: (name, environment, MIFFT_DEBUG_NARROW_TILES)
SWITCHES = (("default", {}, 0), ("narrow_tiles", {}, 1), ("no_rowfirst", {"PYFFT_AMD_NO_SPLIT_ROWFIRST": "1"}, 0),
            ("forced_fused", {"PYFFT_AMD_STRATEGY": "fused"}, 0), ("forced_fused_narrow", {"PYFFT_AMD_STRATEGY": "fused"}, 1),
            ("forced_fused_no_rowfirst", {"PYFFT_AMD_STRATEGY": "fused", "PYFFT_AMD_NO_SPLIT_ROWFIRST": "1"}, 0),
            ("forced_pipelined", {"PYFFT_AMD_STRATEGY": "pipelined"}, 0))
ENV_KEYS = ("PYFFT_AMD_STRATEGY", "PYFFT_AMD_NO_SPLIT_ROWFIRST")


def snapshot(with_chain_class=False):
    from test_host import _FakeContext
    from pyfft_amd import _native as N
    from pyfft_amd.machine import Machine
    from pyfft_amd.plan import FFTPlan
    rows = []
    saved = {k: os.environ.pop(k, None) for k in ENV_KEYS}
    try:
        for sname, env, narrow in SWITCHES:
            for k in ENV_KEYS:
                os.environ.pop(k, None)
            os.environ.update(env)
            N.lib.mifft_debug_set(N.DEBUG_NARROW_TILES, narrow)
            for mname, margs in MACHINES.items():
                if sname != "default" and mname != "full":
                    continue
                mach = Machine(*margs)
                plans = {}
                for shape, dt, batch in cases():
                    key = (shape, dt)
                    if key not in plans:
                        plans[key] = FFTPlan(_FakeContext(mach), shape, dtype=numpy.dtype(dt))
                    st = plans[key]._select_strategy(batch)
                    rows.append([sname, mname, list(shape), dt, batch, list(st)])
                    if with_chain_class:
                        # chains with exactly ONE pass pair are round 5's (the snapshot predates them): the test treats them apart -- as it does
                        # the interleaved 3-D shapes with 256-point axes that got the two pairs of 256^3 late in round 5 (three launches before; round 4 had pairs for the two cubes only)
                        npairs = sum(1 for k in plans[key]._kernels if k.pair_with_next)
                        late = dt in ("complex128", "complex64") and len(shape) == 3 and npairs == 2 and \
                            tuple(shape) not in ((256, 256, 256), (128, 128, 128))
                        # round 6: float32 planes of (16, 16, 128) and 32^3 became ONE launch (dense kernel on 16-byte plane accesses,
                        # csrc/fft_nd2p.hpp; two launches before)
                        planes16 = dt == "float32" and tuple(shape) in ((16, 16, 128), (32, 32, 32)) and len(plans[key]._kernels) == 1
                        rows[-1].append(npairs == 1 or late or planes16)
    finally:
        N.lib.mifft_debug_set(N.DEBUG_NARROW_TILES, 0)
        for k in ENV_KEYS:
            os.environ.pop(k, None)
            if saved[k] is not None:
                os.environ[k] = saved[k]
    return rows


if __name__ == "__main__":
    rows = snapshot()
    import gzip
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "strategy_snapshot.json.gz")
    with open(path, "wb") as raw:
        with gzip.GzipFile(fileobj=raw, mode="wb", mtime=0) as f:
            f.write(("[\n" + ",\n".join(json.dumps(r) for r in rows) + "\n]\n").encode())
    kinds = {}
    for r in rows:
        kinds[r[5][0]] = kinds.get(r[5][0], 0) + 1
    print(len(rows), "rows", kinds)

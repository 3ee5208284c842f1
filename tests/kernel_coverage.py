"""Which kernel instances do default plans select, and which GPU test runs each of them?  (Round 6; test infrastructure, CPU only.)

The library holds several hundred ahead-of-time kernel instances; a plan reaches one through `passes.build_chain` (the chain's
launches: rows, strided passes, one-launch N-D shapes, pass pairs) and through `FFTPlan._select_strategy` (a persistent launch by a
rule of the tuning table, the several-work-groups-per-transform kernel of out-of-place executes).  The rule this module makes
mechanical: EVERY instance some default plan can select must be exercised by a GPU test that the default `pytest -m gpu` run
collects (about half of the pass-pair instances had their only parity test behind a soak switch).

  keys_of(shape, dtype, batch, mode)   the instance keys one test case exercises, computed with the planner itself on the model
                                       of the full part (256 CUs, 8 XCDs, 256 MiB last-level cache) -- no device needed
  universe()                           every key reachable by SOME power-of-two shape of up to 2^24 points in the four dtypes
  collected_cases()                    the (shape, dtype, batch, mode) cases of the default-collected GPU tests, read off the
                                       tests' own parametrize marks through REGISTRY (one adapter per test function)

The reference sweeps its kernels the same way, by shape lists: test/test_errors.py:125-145.
"""
import importlib
import itertools

import numpy

from helpers import FakeContext

GPU_MODULES = ["test_errors_gpu", "test_full_size_gpu", "test_functionality_gpu", "test_random_sweep_gpu", "test_rows_gpu",
               "test_strided_gpu", "test_persistent_gpu", "test_pairs_gpu", "test_nd_gpu", "test_generic_gpu", "test_interop_gpu"]

C64, C128, F32, F64 = numpy.complex64, numpy.complex128, numpy.float32, numpy.float64
ALL_DTYPES = [C64, F32, C128, F64]
_machine = None


def full_machine():
    global _machine
    if _machine is None:
        from pyfft_amd.machine import Machine
        _machine = Machine(256, 8, 4 << 20, 256 << 20)
    return _machine


_plans = {}


def plan_for(shape, dtype):
    """A plan on the model of the full part (no device: tables are uploaded nowhere), cached per (shape, dtype)."""
    from pyfft_amd.plan import FFTPlan
    shape = tuple(int(v) for v in (shape if isinstance(shape, (tuple, list)) else (shape,)))
    key = (shape, numpy.dtype(dtype).name)
    if key not in _plans:
        _plans[key] = FFTPlan(FakeContext(full_machine()), shape, dtype=dtype)
    return _plans[key]


_tables = {}


def fixed_nd_shapes(prec):
    """(x, y, z) of the generated fixed-shape N-D tables csrc/fft_nd2_<prec>_*.hip (tools/gen_nd2_tables.py is their single source)"""
    if "fixed" not in _tables:
        import importlib.util
        import os
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "gen_nd2_tables.py")
        spec = importlib.util.spec_from_file_location("gen_nd2_tables", path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        _tables["fixed"] = {"f32": frozenset(tuple(t) for t in mod.shapes("f32")), "f64": frozenset(tuple(t) for t in mod.shapes("f64"))}
    return _tables["fixed"][prec]


def nd2z_shapes(prec):
    """(x, y, z) with a several-work-groups-per-transform instance, read off the instantiation tables csrc/fft_nd2z_<prec>.hip"""
    if "nd2z" not in _tables:
        import os
        import re
        out = {}
        for name, ctype in (("f32", "float"), ("f64", "double")):
            path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pyfft_amd", "csrc", "fft_nd2z_%s.hip" % name)
            text = "\n".join(l for l in open(path).read().splitlines() if not l.lstrip().startswith("//") and "#define" not in l)
            found = set()
            for m in re.finditer(r"(?:SHAPE[A-Z0-9]*\(|go[x4]?<)\s*%s\s*,\s*(\d+)\s*,\s*(\d+)\s*,\s*(\d+)" % ctype, text):
                found.add(tuple(int(v) for v in m.groups()))
            out[name] = frozenset(found)
        _tables["nd2z"] = out
    return _tables["nd2z"][prec]


def nd2p_shapes(prec):
    """(x, y, z) with a dense split-complex instance on 16-byte plane accesses, read off csrc/fft_nd2p.hip"""
    if "nd2p" not in _tables:
        import os
        import re
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pyfft_amd", "csrc", "fft_nd2p.hip")
        text = open(path).read()
        _tables["nd2p"] = {name: frozenset(tuple(int(v) for v in m.groups())
                                             for m in re.finditer(r"(?:ALL|BIG)\(%s,\s*(\d+),\s*(\d+),\s*(\d+)\)" % ctype, text))
                           for name, ctype in (("f32", "float"), ("f64", "double"))}
    return _tables["nd2p"][prec]


def nd2zp_shapes(prec):
    """(x, y, z) with a several-work-groups-per-transform instance for dense split-complex planes, read off csrc/fft_nd2zp.hip"""
    if "nd2zp" not in _tables:
        import os
        import re
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pyfft_amd", "csrc", "fft_nd2zp.hip")
        text = "\n".join(l for l in open(path).read().splitlines() if not l.lstrip().startswith("//") and "#define" not in l)
        _tables["nd2zp"] = {name: frozenset(tuple(int(v) for v in m.groups())
                                              for m in re.finditer(r"SHAPE[L4]?\(%s,\s*\d,\s*(\d+),\s*(\d+),\s*(\d+)\)" % ctype, text))
                            for name, ctype in (("f32", "float"), ("f64", "double"))}
    return _tables["nd2zp"][prec]


def _nd_key(prec, lay, x, y, z):
    """The instance an N-D pass of this shape runs: the shape's own fixed instance (interleaved: the generated tables; planes: the tiled
    fixed-shape kernel where it takes dense planes), else the ONE run-time-shaped kernel, whose instances differ by the tile's size class."""
    from pyfft_amd import _native as N
    p = N.F64 if prec == "f64" else N.F32
    if lay == "interleaved":
        if (x, y, z) in fixed_nd_shapes(prec) or N.lib.mifft_nd_shape_supported(p, x, y, z, 0) != 0:
            return ("nd_fixed", prec, lay, x, y, z)
    elif (x, y, z) in nd2p_shapes(prec):
        return ("nd_planes16", prec, x, y, z)
    elif x * (8 if prec == "f64" else 4) >= 128 and N.lib.mifft_nd_tiled_supported(p, x, y, z) == 0:
        return ("nd_fixed", prec, lay, x, y, z)
    elif N.lib.mifft_nd_shape_supported(p, x, y, z, 0) != 0:
        return ("nd_fixed", prec, lay, x, y, z)
    n = x * y * z
    return ("nd_generic", prec, lay, "dims%d" % ((x > 1) + (y > 1) + (z > 1)), "log2n=%d" % (n.bit_length() - 1))


def _chain_keys(plan, chain):
    """one key per LAUNCH of a chain"""
    from pyfft_amd import _native as N
    p = plan._params
    prec = "f64" if p.precision == N.F64 else "f32"
    lay = "split" if p.split else "interleaved"
    keys, i = set(), 0
    while i < len(chain):
        k = chain[i]
        if k.pair_with_next:
            k2 = chain[i + 1]
            if k.kind == N.PASS_ROW:
                keys.add(("pairXY", prec, lay, int(k.L), int(k2.L), int(k2.M)))        # (nx, R0, R1): mifft_pair_kernel_supported kind 0
            else:
                keys.add(("pairYZ", prec, lay, int(k.S), int(k.L), int(k2.L)))         # (S0, R1, nz): kind 1
            i += 2
            continue
        if k.kind == N.PASS_ND:
            keys.add(_nd_key(prec, lay, int(k.L), int(k.M), int(k.S)))
        elif k.kind == N.PASS_ROW:
            keys.add(("row", prec, lay, int(k.L)))
        else:
            # a strided pass: its length, whether it carries the inter-pass twiddle (M > 1), and the class of its stride (S == 1: the
            # transposing first pass of a long contiguous axis; S < 16: columns narrower than a tile; else whole 16-column tiles)
            # (fp32 interleaved 256 / 512 points with rows >= 2^16 points apart: the 32-column plain tiles of csrc/fft_col2_f32.hip)
            far = prec == "f32" and lay == "interleaved" and k.L in (256, 512) and k.M == 1 and k.S >= 65536
            keys.add(("col", prec, lay, int(k.L), "twiddled" if k.M > 1 else "plain",
                      "S1" if k.S == 1 else ("S<16" if k.S < 16 else ("S>=65536" if far else "S>=16"))))
        i += 1
    return keys


def _persistent_key(plan):
    from pyfft_amd import _native as N
    p = plan._params
    rule = plan._persistent_rule()
    if rule is None:
        return None
    # (the planes of a big 3-D transform run the 2-D plan's instance: strategy "fused2z")
    return ("persistent", rule["name"], "f64" if p.precision == N.F64 else "f32", "split" if p.split else "interleaved",
            int(p.x), int(p.y), 1 if plan._plane_fused() else int(p.z))


def keys_of(shape, dtype, batch, mode="auto"):
    """Instance keys the test case (shape, dtype, batch) exercises.  mode: "auto" = the plan's own strategy at this batch, in place and
    out of place (what the tests' protocols run); "chain" / "pipelined" = the chain's launches; "fused" = the plan's persistent launch
    whatever the batch (PYFFT_AMD_STRATEGY=fused, or a forced ring)."""
    plan = plan_for(shape, dtype)
    keys = set()
    if mode in ("chain", "pipelined"):
        return _chain_keys(plan, plan._kernels)
    if mode == "fused":
        k = _persistent_key(plan)
        return {k} if k is not None else set()
    strat = plan._select_strategy(int(batch))
    if strat[0] == "fused2z":
        # one persistent 2-D launch over the (y, x) planes -- the 2-D plan's instance; what is new is the planner's route -- then the
        # chain's z launches
        pk = _persistent_key(plan)
        keys.add(pk)
        keys.add(("plane_fused",) + pk[1:4])
        keys |= _chain_keys(plan, plan._kernels[2:])
    elif strat[0] in plan.PERSISTENT:
        keys.add(_persistent_key(plan))
    else:
        keys |= _chain_keys(plan, plan._kernels)
    from pyfft_amd import _native as N
    p = plan._params
    prec = "f64" if p.precision == N.F64 else "f32"
    if plan._runs_oop_nd(int(batch)):
        keys.add(("nd_oop", prec + ("_planes" if p.split else ""), int(p.x), int(p.y), int(p.z)))
    # a one-launch plan of split-complex planes with a several-work-groups instance (csrc/fft_nd2zp.hip): its out-of-place executes
    if p.split and len(plan._kernels) == 1 and plan._kernels[0].kind == N.PASS_ND and (int(p.x), int(p.y), int(p.z)) in nd2zp_shapes(prec):
        keys.add(("nd2zp", prec, int(p.x), int(p.y), int(p.z)))
    # a one-launch plan of a shape with a several-work-groups-per-transform instance: its OUT-OF-PLACE executes take that instance inside
    # the library (csrc/mifft_runtime.cpp launch_nd; small launches only for some shapes -- the tests assert which)
    if not p.split and len(plan._kernels) == 1 and plan._kernels[0].kind == N.PASS_ND and (int(p.x), int(p.y), int(p.z)) in nd2z_shapes(prec):
        keys.add(("nd2z", prec, int(p.x), int(p.y), int(p.z)))
    return keys


# pair kernels of the library that only a development switch selects (MIFFT_DEBUG_PAIR = 2: the other y split of 256^3, interleaved fp64);
# tests/test_pairs_gpu.py::test_pass_pairs_256_cubed_alternative_split runs it
DEV_ONLY_PAIR_KEYS = {("pairXY", "f64", "interleaved", 256, 64, 4), ("pairYZ", "f64", "interleaved", 16384, 4, 256)}

MAX_LOG2_POINTS = 24          # shapes of up to 2^24 points: 4096 x 4096, 256^3, 2^24 (128 MiB fp32 / 256 MiB fp64 per transform)


def universe(max_log=MAX_LOG2_POINTS):
    """{key: an example (shape, dtype, batch)} over every power-of-two shape of up to 2^max_log points (x contiguous; numpy shapes
    (z, y, x)), the four dtypes, and batches of 512 MiB, 2 GiB and 8 GiB per side -- where the planner's default choices live."""
    out = {}
    for lx in range(0, max_log + 1):
        for ly in range(0, max_log + 1 - lx):
            for lz in range(0, max_log + 1 - lx - ly):
                if lx + ly + lz < 1:
                    continue
                if (ly == 0 and lz > 0) or (lx == 0 and (ly > 0 or lz > 0)):
                    continue            # (unit axes in front of longer ones are the same plans as the lower-rank shape: tests cover them by name)
                shape = (1 << lx,) if ly == 0 else ((1 << ly, 1 << lx) if lz == 0 else (1 << lz, 1 << ly, 1 << lx))
                for dtype in ALL_DTYPES:
                    item = (1 << (lx + ly + lz)) * (8 if dtype in (C64, F32) else 16)
                    for side in (512 << 20, 2 << 30, 8 << 30):
                        batch = max(1, side // item)
                        for k in keys_of(shape, dtype, batch):
                            if k is not None and k not in out:
                                out[k] = (shape, numpy.dtype(dtype).name, batch)
    return out


# ---- the default-collected GPU cases, read off the tests' parametrize marks ----------------------------------------------------------
def params_of(fn):
    """[{argname: value}] of a test function's (stacked) parametrize marks"""
    combos = [{}]
    for m in getattr(fn, "pytestmark", []):
        if m.name != "parametrize":
            continue
        names = [n.strip() for n in m.args[0].split(",")] if isinstance(m.args[0], str) else list(m.args[0])
        new = []
        for c in combos:
            for v in m.args[1]:
                vals = v.values if type(v).__name__ == "ParameterSet" else v
                if len(names) == 1:
                    vals = (vals,)
                d = dict(c)
                d.update(zip(names, vals))
                new.append(d)
        combos = new
    return combos


def _prod(shape):
    return int(numpy.prod(shape))


def _c(shape, dtype, batch, mode="auto"):
    return (tuple(shape), dtype, int(batch), mode)


def _fixed_nd(p):
    x, y, z = p["xyz"]
    shape = (y, x) if z == 1 else (z, y, x)
    n = x * y * z
    tile = 4096 if p["dtype"] == C64 else 2048
    return [_c(shape, p["dtype"], 3 if n >= tile else (tile // n) * 2 + 1)]


def _is_double(dt):
    return numpy.dtype(dt) in (numpy.dtype(C128), numpy.dtype(F64))


def _split(dt):
    return numpy.dtype(dt).kind == "f"


# adapter(params) -> [(shape, dtype, batch, mode)]; None = the test builds no default power-of-two plan worth counting (API checks,
# direct C-ABI launches, opt-in extensions, development strategies of `make DEV=1` builds)
REGISTRY = {
    "test_errors_gpu": {
        "test_errors_batch_1_and_3": lambda p: [_c(p["shape"], p["dtype"], b) for b in (1, 3) if _prod(p["shape"]) * b <= (1 << 19)],
        "test_errors_batched": lambda p: [_c(p["shape"], p["dtype"], p["batch"]), _c(p["shape"], p["dtype"], p["batch"] - 1 if p["batch"] > 1 else 5)],
        "test_register_edged_rows": lambda p: [_c((p["n"],), p["dtype"], 37)] + ([_c((128, p["n"]), p["dtype"], 3)] if p["n"] <= 4096 else []),
        "test_fixed_shape_nd_kernels": _fixed_nd,
        "test_fixed_shape_nd_planes": lambda p: [_c(p["shape"], p["dtype"], p["batch"])],
        "test_fp64_two_phase_col_512": lambda p: [_c(p["shape"], p["dtype"], p["batch"])],
        "test_unit_axes": lambda p: [_c(p["shape"], p["dtype"], 3)],
        "test_errors_large_1d": lambda p: [_c((1 << 20,), p["dtype"], 2)],
        "test_errors_config3_shape": lambda p: [_c((1024, 1024), C64, 2)],
        "test_errors_config4_shape_reduced": lambda p: [_c((64, 64, 64), p["dtype"], 1), _c((256, 16, 16), p["dtype"], 2)],
        "test_fused_two_pass_kernel": lambda p: [_c((p["n"],), C64, p["batch"], "fused"), _c((p["n"],), C64, p["batch"], "chain")],
        "test_fused_2d_1024": lambda p: [_c((1024, 1024), p["dtype"], 29 if _is_double(p["dtype"]) else 57, "fused"),
                                         _c((1024, 1024), p["dtype"], 29 if _is_double(p["dtype"]) else 57, "chain")],
        "test_fused_2d_other_squares": lambda p: [_c((p["side"], p["side"]), C64, p["batch"])],
        "test_fused_two_pass_kernel_fp64": lambda p: [_c((1 << 20,), C128, 31, "fused"), _c((1 << 20,), C128, 31, "chain"), _c((1 << 20,), F64, 31, "fused")],
        "test_pipelined_chunks": lambda p: [_c(p["shape"], C64, p["batch"], "pipelined")],
        "test_split_plane_strategies": lambda p: [_c((p["n"],), F32, p["batch"], p["strat"]), _c((p["n"],), F32, p["batch"], "chain")],
        "test_golden_fixture_through_hip": lambda p: [_c(p["entry"]["shape"], numpy.dtype(p["entry"]["dtype"]).type, p["entry"]["batch"])],
        "test_reference_error_grid": lambda p: [_c(p["shape"], p["dtype"], p["batch"])],
    },
    "test_full_size_gpu": {
        "test_config2_full": lambda p: [_c((1 << 20,), C64, 4096)],
        "test_config3_full": lambda p: [_c((1024, 1024), C64, 512)],
        "test_config4_full": lambda p: [_c((256, 256, 256), p["dtype"], 64)],
        "test_config5_per_gpu_chunk": lambda p: [_c((1 << 22,), C64, 256)],
        "test_linearity_large": lambda p: [_c((1 << 20,), C64, 4)],
    },
    "test_functionality_gpu": None,      # the reference's API checks (test/test_functionality.py) on tiny shapes
    "test_random_sweep_gpu": {
        "test_random_case": lambda p: [_c(p["case"][0], p["case"][2], p["case"][1])],
    },
    "test_rows_gpu": {
        "test_small_n_wave_path_parity": lambda p: [_c((p["n"],), p["dtype"], b) for b in (1, 3, 63, 64, 65, 1000)],
        "test_wave_kernel_matches_lds_kernel": None,
        "test_16x16_wave_plane_kernel": None,
        "test_small_launch_write_through_is_bit_identical": lambda p: [_c(p["shape"], p["dtype"], p["batch"])],
        "test_write_through_rule_stops_at_128_mib": None,
    },
    "test_strided_gpu": {
        "test_l2048_strided_pass": lambda p: [_c(p["shape"], p["dtype"], p["batch"])],
        "test_l1024_fp64_strided_pass": lambda p: [_c(p["shape"], p["dtype"], p["batch"])],
        "test_l2048_fp64_strided_pass": lambda p: [_c(p["shape"], C128, p["batch"])],
        "test_l2048_fp64_fallback_kernel": None,
        "test_register_only_short_strided_pass": None,
        "test_long_strided_axes": lambda p: [_c(p["shape"], p["dtype"], 1 if _prod(p["shape"]) * numpy.dtype(p["dtype"]).itemsize * (2 if _split(p["dtype"]) else 1) >= (64 << 20) else 2)],
    },
    "test_persistent_gpu": {
        "test_async_error_mailbox": lambda p: [_c((1 << 20,), C64, 64)],
        "test_xcd2_strategy_bit_identical_and_in_place": None,
        "test_error_mailbox_keeps_the_oldest_word_when_the_ring_is_full": None,
        "test_fused_pair_cube_128": lambda p: [_c((128, 128, 128), p["dtype"], p["batch"], "fused"), _c((128, 128, 128), p["dtype"], p["batch"], "chain")],
        "test_per_xcd_lists": None,
        "test_fused_ring_rule_2_19": lambda p: [_c((1 << 19,), C64, 130), _c((1 << 19,), C64, 130, "chain")],
        "test_sequential_single_launch_of_tiny_batches": None,
        "test_alternating_counter_sets_and_memset_form_agree": lambda p: [_c((1 << 20,), C64, 70)],
        "test_fused_2d_rectangles": lambda p: [_c(p["shape"], C64, p["batch"], "fused" if tuple(p["shape"]) == (512, 2048) else "auto"),
                                               _c(p["shape"], C64, p["batch"], "chain")],
        "test_fused_long_fp64": lambda p: [_c((p["n"],), C128, p["batch"], "fused"), _c((p["n"],), C128, p["batch"], "chain")],
        "test_fused_mid_sizes_fp64": lambda p: [_c((p["n"],), C128, p["batch"], "fused"), _c((p["n"],), C128, p["batch"], "chain")],
        "test_fused_2d_fp64_512_sides": lambda p: [_c(p["shape"], C128, p["batch"], "fused"), _c(p["shape"], C128, p["batch"], "chain")],
        "test_wide_tiles_fp32_mid_sizes": lambda p: [_c((p["n"],), C64, p["batch"]), _c((p["n"],), C64, p["batch"], "chain")],
        "test_fused_2d_256_sides": lambda p: ([] if (_is_double(p["dtype"]) and max(p["shape"]) > 512) else
                                              [_c(p["shape"], p["dtype"], p["batch"] // 2 + 1 if _is_double(p["dtype"]) else p["batch"]),
                                               _c(p["shape"], p["dtype"], p["batch"], "chain")]),
        "test_split_planes_on_per_xcd_lists": lambda p: [_c((p["n"],), F32, p["batch"]), _c((p["n"],), F32, p["batch"], "chain")],
        "test_fused_pair_small_axes": lambda p: [_c(p["shape"], p["dtype"], p["batch"] // 2 + 1 if _is_double(p["dtype"]) else p["batch"]),
                                                 _c(p["shape"], p["dtype"], p["batch"], "chain")],
        "test_fused_pair_split_planes": lambda p: [_c(p["shape"], p["rdtype"], p["batch"]), _c(p["shape"], p["rdtype"], p["batch"], "chain")],
        "test_plane_fused_3d": lambda p: [_c(p["shape"], p["dtype"], p["batch"]), _c(p["shape"], p["dtype"], p["batch"], "chain")],
        "test_fused_2d_split_row_first": lambda p: [_c(p["shape"], F32, p["batch"], "fused" if tuple(p["shape"]) == (256, 256) else "auto"),
                                                    _c(p["shape"], F32, p["batch"], "chain")],
        "test_fused_split_planes_fp64": lambda p: [_c(p["shape"], F64, p["batch"]), _c(p["shape"], F64, p["batch"], "chain")],
        "test_config5_per_gpu_share": lambda p: [_c((1 << 22,), C64, 256)],
        "test_config5_share_as_one_execute": lambda p: [_c((1 << 22,), C64, 8192)],
        "test_config5_share_every_output_bit_identical_to_its_period_mate": lambda p: [_c((1 << 22,), C64, 8192)],
        "test_direct_abi_two_set_launch_refuses_capture_and_null_error_word": lambda p: [_c((1 << 18,), C64, 160)],
    },
    "test_pairs_gpu": {
        "test_pass_pairs_256_cubed": lambda p: [_c((256, 256, 256), p["dtype"], 1)],
        "test_pass_pairs_256_cubed_alternative_split": None,         # (a development A/B instance: DEV_ONLY_PAIR_KEYS below)
        "test_pair_chains_two_launches": lambda p: [_c(p["shape"], p["dtype"], p["batch"])],
        "test_pair_chains_pipelined_chunks": lambda p: [_c(p["shape"], p["dtype"], p["batch"])],
        "test_pass_pairs_for_256_point_rows": lambda p: [_c(p["shape"], p["dtype"], p["batch"])],
    },
    "test_nd_gpu": {
        "test_two_work_groups_per_transform_nd": "nd2z",        # (the shapes are a list inside the module: see collected_cases)
        "test_four_work_groups_per_transform_out_of_place": lambda p: [
            _c(p["shape"], p["dtype"], (261 * 65536 * 8) // (_prod(p["shape"]) * numpy.dtype(p["dtype"]).itemsize))],
        "test_two_work_groups_per_transform_split_planes": lambda p: [_c(p["shape"], F32, 13 if p["small"] else (261 * 65536 * 8) // (_prod(p["shape"]) * 8))],
        "test_tiny_nd_shapes_small_launches": lambda p: [_c(p["shape"], p["dtype"], p["batch"])],
        "test_tiny_nd_shapes_big_launches": lambda p: [_c(p["shape"], p["dtype"], p["batch"])],
        "test_run_time_shaped_nd_kernel_size_classes": lambda p: [_c(p["shape"], p["dtype"], 5 if _prod(p["shape"]) >= 4096 else 4099 // _prod(p["shape"]))],
        "test_dense_split_planes_on_the_tiled_fixed_kernels": lambda p: [_c(p["shape"], p["dtype"], p["batch"])],
        "test_dense_split_planes_16_byte_accesses": lambda p: [_c(p["shape"], p["dtype"], 1)],
    },
    "test_generic_gpu": None,            # opt-in extensions (any_size / parent_shape): their inner power-of-two plans are ordinary plans
    "test_interop_gpu": {
        "test_reference_quick_start_literally": lambda p: [_c((1 << 22,), C64, 1)],
        "test_default_plan_orders_against_torch_producer": lambda p: [_c((1 << 20,), C64, 64)],
        "test_context_device_must_be_current": None,
        "test_bench_nccl_path_at_world_size_one": lambda p: [_c((1 << 20,), C64, 128)],
        "test_plan_close_releases_and_plan_stays_usable": lambda p: [_c((1 << 16,), C64, 64)],
        "test_two_ranks_share_one_gpu_sharded_path": None,
        "test_hipfft_and_libmifft_agree_on_the_comparator_shapes": None,
        "test_perf_table_quick_uses_the_reference_formula": None,
        "test_plan_following_two_torch_streams_without_host_sync": lambda p: [_c((1 << 18,), C64, 160)],
        "test_plan_for_a_device_given_by_index": lambda p: [_c((4096,), C64, 4)],
        "test_device_properties_describe_the_memory_system": None,
        "test_plan_with_stream_and_context_index": lambda p: [_c((8192,), C64, 3)],
        "test_eight_ranks_share_one_gpu": lambda p: [_c((1 << 20,), C64, 40)],
        "test_captured_execute_replays_bit_identically": lambda p: [_c(p["shape"], p["dtype"], p["batch"])],
        "test_plans_of_four_host_threads_run_side_by_side": "threads",
        "test_torch_cuda_graph_around_execute": lambda p: [_c((1 << 18,), C64, 160)],
        "test_sharded_plan_two_shards_on_one_device": lambda p: [_c(p["shape"], p["dtype"], p["batch"] // 2)],
        "test_bench_single_process_four_shards_share_one_gpu": lambda p: [_c((1 << 20,), C64, 40)],
    },
}


def test_functions(modname):
    mod = importlib.import_module(modname)
    return mod, sorted(n for n in dir(mod) if n.startswith("test_") and callable(getattr(mod, n)))


def collected_cases():
    """[(module, test, (shape, dtype, batch, mode))] of every default-collected GPU test; raises if a test function of a GPU module
    has no entry in REGISTRY (a new test must say what it covers)."""
    import os
    assert not os.environ.get("PYFFT_AMD_SWEEP"), "the coverage rule is about the DEFAULT collection: unset PYFFT_AMD_SWEEP"
    out = []
    for modname in GPU_MODULES:
        mod, fns = test_functions(modname)
        reg = REGISTRY[modname]
        if reg is None:
            continue
        missing = [f for f in fns if f not in reg]
        stale = [f for f in reg if f not in fns]
        if missing or stale:
            raise AssertionError("tests/kernel_coverage.py REGISTRY[%r]: no entry for %r, stale entries %r" % (modname, missing, stale))
        for f in fns:
            ad = reg[f]
            if ad is None:
                continue
            if ad == "nd2z":
                cases = [_c(sh, dt, 11) for dt in (C64, C128) for sh in mod.ND2Z_SHAPES[dt]]
            elif ad == "threads":
                cases = [_c(sh, dt, b) for sh, b, dt, _ in mod.THREAD_CASES]
            else:
                cases = list(itertools.chain.from_iterable(ad(p) for p in params_of(getattr(mod, f))))
            out += [(modname, f, c) for c in cases]
    return out


def covered_keys(cases=None):
    """{key: (module, test)} over the collected cases"""
    out = {}
    for modname, f, (shape, dtype, batch, mode) in (collected_cases() if cases is None else cases):
        for k in keys_of(shape, dtype, batch, mode):
            if k is not None:
                out.setdefault(k, (modname, f))
    return out

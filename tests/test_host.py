"""CPU tests of the host side: the C-ABI library loads and exports every declared symbol, the
gfx950 pass decomposition is consistent with the oracle's pass algebra, and the Python API raises
the reference's errors.  No compute calls (no GPU here)."""
import os
import re

import numpy
import pytest

import pyfft_oracle as oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    import ctypes
    from pyfft_amd import _native as N
    header = open(os.path.join(ROOT, "include", "mifft.h")).read()
    declared = set(re.findall(r"\b(mifft_[a-z0-9_]+)\s*\(", header))
    declared -= {"mifft_pass", "mifft_device_props"}
    assert len(declared) >= 25
    lib = ctypes.CDLL(N.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), "libmifft.so does not export %s" % name
    assert declared == set(N.PROTOTYPES.keys())
    assert N.lib.mifft_abi_version() == N.ABI_VERSION


def test_struct_layout_matches_header():
    import ctypes
    from pyfft_amd import _native as N
    assert ctypes.sizeof(N.MifftPass) == 112
    assert N.MifftPass.scale.offset == 64 and N.MifftPass.tw_L.offset == 72 and N.MifftPass.src.offset == 100


def test_supported_lengths():
    from pyfft_amd import _native as N
    from pyfft_amd import passes as P
    for prec in (N.F32, N.F64):
        assert P.row_max(prec) >= 2048 and P.col_max(prec) >= 256
        L = 2
        while L <= P.row_max(prec):
            assert N.lib.mifft_pass_supported(N.PASS_ROW, prec, L, 0) == 0
            L *= 2
        L = 2
        while L <= P.col_max(prec):
            assert N.lib.mifft_pass_supported(N.PASS_COL, prec, L, 0) == 0
            L *= 2
        assert N.lib.mifft_pass_supported(N.PASS_ROW, prec, 3, 0) == N.E_UNSUPPORTED
        assert N.lib.mifft_pass_supported(N.PASS_COL, prec, 1 << 20, 0) == N.E_UNSUPPORTED
    # the longest rows exist for planes too (second batch of round 4), so a split-complex 1-D plan of that length is one pass
    assert P.row_max(N.F32) == P.row_max(N.F32, interleaved=True) == 32768 and P.row_max(N.F64) == P.row_max(N.F64, interleaved=True) == 16384
    assert [k.kind for k in P.build_chain(32768, 1, 1, N.F32, interleaved=False)] == [N.PASS_ROW]
    # single-launch N-D shapes beyond the run-time-shaped kernel's tile: interleaved only, or (a few, with long x rows) planes on both sides
    big = N.lib.mifft_nd_max_points_for(N.F64)
    assert 128 * 128 > big and N.lib.mifft_nd_shape_supported(N.F64, 128, 128, 1, 0) == N.E_UNSUPPORTED
    assert N.lib.mifft_nd_shape_supported(N.F64, 128, 128, 1, N.VARIANT_INTERLEAVED_ONLY) == 0
    assert N.lib.mifft_nd_shape_supported(N.F64, 128, 128, 1, N.VARIANT_SPLIT_ONLY) == 0
    # (fp32 planes beyond the run-time-shaped kernel's tile: the dense 16-byte kernel of round 6 has (16, 16, 128); a shape without such an
    # instance stays two passes)
    assert N.lib.mifft_nd_shape_supported(N.F32, 128, 16, 16, N.VARIANT_SPLIT_ONLY) == 0
    assert N.lib.mifft_nd_shape_supported(N.F32, 64, 64, 8, N.VARIANT_SPLIT_ONLY) == N.E_UNSUPPORTED
    assert [k.kind for k in P.build_chain(128, 128, 1, N.F64, interleaved=False)] == [N.PASS_ND]
    assert len(P.build_chain(64, 64, 8, N.F32, interleaved=False)) == 2


def test_launch_rejects_bad_descriptors_without_touching_the_gpu():
    import ctypes
    from pyfft_amd import _native as N
    p = N.MifftPass()
    p.kind, p.precision, p.layout, p.L, p.M, p.S, p.outer = N.PASS_ROW, N.F32, N.INTERLEAVED, 12, 1, 1, 1
    p.tw_L = 16
    assert N.lib.mifft_launch_pass(ctypes.byref(p), 16, None, 16, None, None) == N.E_INVALID
    assert "power of two" in N.last_error()
    p.L = 16
    assert N.lib.mifft_launch_pass(ctypes.byref(p), 8, None, 16, None, None) == N.E_INVALID   # misaligned
    p.kind, p.M, p.S = N.PASS_COL, 4, 1
    assert N.lib.mifft_launch_pass(ctypes.byref(p), 16, None, 32, None, None) == N.E_INVALID  # tables missing
    with pytest.raises(ValueError):
        N.check(N.E_INVALID, "x")
    with pytest.raises(RuntimeError):
        N.check(N.E_UNSUPPORTED, "x")


SHAPES = [(2, 1, 1), (1024, 1, 1), (4096, 1, 1), (8192, 1, 1), (1 << 16, 1, 1), (1 << 20, 1, 1), (1 << 22, 1, 1),
          (1 << 24, 1, 1), (16, 16, 1), (1024, 1024, 1), (4, 2048, 1), (2048, 8, 1), (256, 256, 256), (16, 16, 16),
          (64, 8, 8), (2, 4, 2), (1, 16, 1), (1, 1, 64), (1, 8, 8), (16, 1, 8), (128, 128, 128), (128, 32, 32), (64, 16, 2048)]


@pytest.mark.parametrize("xyz", SHAPES, ids=str)
def test_gfx950_chain_is_a_valid_factorisation_and_matches_numpy(xyz):
    """Run the build's own chain (passes.build_chain) through the oracle's pass algebra
    (global_pass) on a small batch: the decomposition + the reference formula == numpy.fft."""
    from pyfft_amd import _native as N
    from pyfft_amd import passes as P
    x, y, z = xyz
    chain = P.build_chain(x, y, z, N.F32)
    if len(chain) == 1 and chain[0].kind == N.PASS_ND:
        # whole small N-D transform in one launch: the descriptor just carries the three axis lengths
        k = chain[0]
        assert (k.L, k.M, k.S) == (x, y, z) and k.in_place_possible and k.outer_stride == x * y * z
        assert x * y * z <= N.lib.mifft_nd_max_points_for(N.F32)
        return
    if chain and chain[0].kind == N.PASS_ND:
        # (y, x) planes in LDS, then the z chain
        k = chain[0]
        assert (k.L, k.M, k.S) == (x, y, 1) and k.outer_per_batch == z and k.outer_stride == x * y
        assert all(c.kind == N.PASS_COL and c.axis == P.Z_DIRECTION for c in chain[1:]) and len(chain) >= 2
        prod = 1
        for c in chain[1:]:
            prod *= c.L
        assert prod == z
        return
    per_axis = {}
    for k in chain:
        per_axis[k.axis] = per_axis.get(k.axis, 1) * k.L
        assert k.L >= 2 and (k.kind == N.PASS_ROW) == (k.M * k.S == 1)
        assert k.in_place_possible == (k.M == 1)
        assert k.L <= (P.row_max(N.F32) if k.kind == N.PASS_ROW else P.col_max(N.F32))
    for axis, n in zip((0, 1, 2), (x, y, z)):
        assert per_axis.get(axis, 1) == n
    if x * y * z > (1 << 16):
        return
    batch = 2
    rng = numpy.random.default_rng(3)
    data = rng.standard_normal((batch, z, y, x)) + 1j * rng.standard_normal((batch, z, y, x))
    cur = data.reshape(-1).copy()
    for k in chain:
        per = k.outer_stride
        assert cur.size == k.outer_per_batch * batch * per
        cur = oracle.global_pass(cur, k.L, k.M, k.S, -1).reshape(-1)
    ref = numpy.fft.fftn(data, axes=(1, 2, 3)).reshape(-1)
    assert numpy.abs(cur - ref).max() < 1e-9 * max(1.0, numpy.abs(ref).max())


@pytest.mark.parametrize("xyz", SHAPES, ids=str)
def test_buffer_schedule_contract(xyz):
    from pyfft_amd import _native as N
    from pyfft_amd import passes as P
    chain = P.build_chain(*xyz, N.F32)
    for inplace in (False, True):
        temp, sched = P.buffer_schedule(chain, inplace)
        assert len(sched) == len(chain) and sched[-1][1] == 1
        loc = 1 if inplace else 0
        for k, (r, w) in zip(chain, sched):
            if inplace and r == 0:
                r = 1              # in place: data_out aliases data_in
            assert r == loc
            if not inplace:
                assert w != 0                 # out-of-place never writes data_in
            if not k.in_place_possible:
                assert r != w
            if 2 in (r, w):
                assert temp
            loc = w
    # same rules as the oracle's restatement of plan.py:200-248
    class K(object):
        def __init__(self, ip):
            self.in_place_possible = ip
    for flags in [(True,), (False, True), (False, False, True), (True, False, True), (True, True, True, False, True)]:
        ks = [K(f) for f in flags]
        for inplace in (False, True):
            assert P.buffer_schedule(ks, inplace) == oracle.buffer_schedule(ks, inplace)


def test_plan_argument_errors_without_device():
    """test_functionality.py:129-139: ValueError for bad size / dtype / shape, raised before any
    device is touched; with valid arguments and no GPU the factory fails loudly (no CPU fallback)."""
    import pyfft_amd.hip as hip
    with pytest.raises(ValueError):
        hip.Plan((17,), dtype=numpy.complex64)
    with pytest.raises(ValueError):
        hip.Plan((16,), dtype=numpy.int32)
    with pytest.raises(ValueError):
        hip.Plan((16, 16, 16, 16), dtype=numpy.complex64)
    with pytest.raises(ValueError):
        hip.Plan("16", dtype=numpy.complex64)
    with pytest.raises(ValueError):
        hip.Plan((16, 24), dtype=numpy.complex64)     # per-axis check (the reference only checks the product)
    if hip.device_count() == 0:
        with pytest.raises(RuntimeError):
            hip.Plan((16,), dtype=numpy.complex64)


def test_version_is_tuple_of_ints():
    import pyfft_amd
    assert isinstance(pyfft_amd.VERSION, tuple) and all(isinstance(v, int) for v in pyfft_amd.VERSION)


def test_device_pointer_extraction():
    import ctypes
    import pyfft_amd.hip as hip

    class WithGpudata(object):
        gpudata = 4096

    class WithIface(object):
        __cuda_array_interface__ = {"data": (8192, False)}

    assert hip.device_pointer(1234) == 1234
    assert hip.device_pointer(WithGpudata()) == 4096
    assert hip.device_pointer(WithIface()) == 8192
    assert hip.device_pointer(ctypes.c_void_p(64)) == 64
    assert hip.device_pointer(None) is None
    torch = pytest.importorskip("torch")
    t = torch.zeros(4)
    assert hip.device_pointer(t) == t.data_ptr()
    with pytest.raises(TypeError):
        hip.device_pointer("nope")


def test_kernel_model_matches_oracle():
    """The numpy model of fft_tile.hpp's index algebra (tests/kernel_model.py) against the oracle."""
    import kernel_model as km
    rng = numpy.random.default_rng(0)

    def rnd(n):
        return rng.standard_normal(n) + 1j * rng.standard_normal(n)
    for L, W, NT, rad in [(16, 8, 8, [16]), (64, 4, 16, [8, 8]), (1024, 1, 64, [16, 16, 4]), (2, 32, 4, [2])]:
        x = rnd(5 * L)
        y = km.run_pass(x, True, L, 1, 1, 5, L, W, NT, rad)
        assert numpy.abs(y - numpy.fft.fft(x.reshape(5, L), axis=1).ravel()).max() < 1e-9
    for L, M, S, W, NT, rad, outer in [(16, 1, 8, 8, 8, [16], 3), (16, 4, 1, 8, 8, [16], 3), (64, 2, 4, 4, 16, [8, 8], 2),
                                       (32, 8, 1, 4, 8, [8, 4], 1), (128, 4, 2, 4, 32, [16, 8], 1)]:
        x = rnd(outer * L * M * S)
        y = km.run_pass(x, False, L, M, S, outer, L * M * S, W, NT, rad)
        assert numpy.abs(y - oracle.global_pass(x, L, M, S, -1)).max() < 1e-9


def test_via_temp_schedule_contract():
    """fp32 split-plane plans whose passes could all run in place detour through the (interleaved) temp:
    in -> temp, temp -> temp ..., temp -> out.  Same contract: out-of-place never writes data_in, the result
    lands in data_out / data_in."""
    from pyfft_amd import _native as N
    from pyfft_amd import passes as P
    for xyz in [(1024, 1024, 1), (256, 256, 256), (64, 64, 512)]:
        chain = P.build_chain(*xyz, N.F32)
        assert all(k.in_place_possible for k in chain) and len(chain) >= 2
        for inplace in (False, True):
            temp, sched = P.buffer_schedule(chain, inplace, via_temp=True)
            assert temp and len(sched) == len(chain)
            assert sched[0] == ((1 if inplace else 0), 2) and sched[-1] == (2, 1)
            assert all(sd == (2, 2) for sd in sched[1:-1])
            assert all(w != 0 for _, w in sched)
    # single-pass plans and plans that need the temp anyway are unchanged
    for xyz in [(1024, 1, 1), (1 << 20, 1, 1), (16, 16, 1)]:
        chain = P.build_chain(*xyz, N.F32)
        for inplace in (False, True):
            assert P.buffer_schedule(chain, inplace, via_temp=True) == P.buffer_schedule(chain, inplace)


def test_integration_stub_matches_the_bindings():
    """The reference-side ctypes stub shown in INTEGRATION.md declares struct mifft_pass field for field as
    pyfft_amd/_native.py does (which test_struct_layout_matches_header ties to include/mifft.h)."""
    import re
    from pyfft_amd import _native as N
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = text[text.index("class MifftPass(ctypes.Structure):"):]
    block = block[:block.index("_lib.mifft_last_error")]
    fields = re.findall(r'\("(\w+)", ctypes\.(c_\w+)\)', block)
    import ctypes
    got = [(name, getattr(ctypes, ct)) for name, ct in fields]      # (c_int32 is an alias of c_int: compare the types)
    assert got == list(N.MifftPass._fields_)


def test_bluestein_tables_reproduce_the_dft_on_the_host():
    """pyfft_amd/generic.py's chirp and filter tables, pushed through numpy FFTs the way the plan pushes them through the
    power-of-two kernels, give the DFT of any length (the device only adds rounding)."""
    from pyfft_amd import generic
    rng = numpy.random.default_rng(5)
    for n in (3, 5, 12, 100, 1023):
        m = 1 << int(numpy.ceil(numpy.log2(2 * n - 1)))
        c = generic._chirp(n, numpy.complex128)
        assert numpy.allclose(c, numpy.exp(-1j * numpy.pi * (numpy.arange(n) ** 2 % (2 * n)) / n), atol=1e-15)
        b = numpy.zeros(m, numpy.complex128)
        b[:n] = numpy.conj(c)
        b[m - n + 1:] = numpy.conj(c[1:][::-1])
        x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        a = numpy.zeros(m, numpy.complex128)
        a[:n] = x * c
        y = numpy.fft.ifft(numpy.fft.fft(a) * numpy.fft.fft(b))
        assert numpy.allclose(y[:n] * c, numpy.fft.fft(x), rtol=0, atol=1e-10 * numpy.abs(x).sum())


def test_generic_plan_index_spaces():
    """The gather / scatter index space of a tiled batch addresses every element of the parent arrays exactly once."""
    from pyfft_amd.generic import GenericFFTPlan
    plan = GenericFFTPlan.__new__(GenericFFTPlan)
    plan._xyz, plan._parent, plan._counts = (4, 2, 2), (8, 6, 4), (2, 3, 2)
    dims, user, work = plan._user_dims_strides(3)
    idx = numpy.indices(dims[::-1]).reshape(6, -1)[::-1]          # idx[d] = index along dims[d]
    uoff = sum(i * s for i, s in zip(idx, user))
    woff = sum(i * s for i, s in zip(idx, work))
    total = 3 * 8 * 6 * 4
    assert sorted(uoff.tolist()) == list(range(total)) and sorted(woff.tolist()) == list(range(total))
    # a work-array tile is one dense (z, y, x) block of the parent
    par = numpy.arange(total).reshape(3, 4, 6, 8)
    w = numpy.empty(total, numpy.int64)
    w[woff] = par.ravel()[uoff]
    tiles = w.reshape(3 * 2, 3, 2, 2, 2, 4)                      # [item*cz][cy][cx][z][y][x]
    assert numpy.array_equal(tiles[1 * 2 + 1, 2, 1], par[1, 2:4, 4:6, 4:8])


def test_plan_device_guard_switches_and_restores():
    """plan.on_plan_device (Plan(context=i), cuda.py:121-128): a guarded context is activated around the call and the caller's
    device restored afterwards, also when the call raises; an unguarded one is left alone."""
    from pyfft_amd.plan import on_plan_device

    class Ctx(object):
        def __init__(self, guard):
            self._guard = guard
            self.log = []

        def activate(self):
            self.log.append("activate")
            return 3

        def restore(self, prev):
            self.log.append(("restore", prev))

    class P(object):
        def __init__(self, guard):
            self._context = Ctx(guard)

        @on_plan_device
        def work(self, x, fail=False):
            """doc"""
            self._context.log.append("work")
            if fail:
                raise RuntimeError("boom")
            return x + 1

    p = P(True)
    assert p.work(1) == 2 and p._context.log == ["activate", "work", ("restore", 3)]
    p = P(True)
    try:
        p.work(1, fail=True)
    except RuntimeError:
        pass
    assert p._context.log == ["activate", "work", ("restore", 3)]
    p = P(False)
    assert p.work(5) == 6 and p._context.log == ["work"]
    assert P.work.__name__ == "work" and P.work.__doc__ == "doc"


def test_pass_pair_chain_algebra_and_schedule():
    """passes.pair_chain (the two pass pairs of csrc/fft_pair.hpp): the four passes pushed through the oracle's pass algebra
    reproduce numpy.fft.fftn on small shapes; the buffer schedule treats a pair as one launch (both descriptors carry the
    unit's src / dst, the (ROW x, COL y) pair never runs in place); the library reports the split it has kernels for."""
    from pyfft_amd import _native as N
    from pyfft_amd import passes as P
    rng = numpy.random.default_rng(8)
    for (x, y, z, r0) in ((8, 16, 4, 4), (4, 8, 8, 2), (16, 4, 2, 2)):
        chain = P.pair_chain(x, y, z, r0)
        assert [k.pair_with_next for k in chain] == [True, False, True, False]
        batch = 2
        data = rng.standard_normal((batch, z, y, x)) + 1j * rng.standard_normal((batch, z, y, x))
        cur = data.reshape(-1).copy()
        for k in chain:
            assert cur.size == k.outer_per_batch * batch * k.outer_stride
            cur = oracle.global_pass(cur, k.L, k.M, k.S, -1).reshape(-1)
        ref = numpy.fft.fftn(data, axes=(1, 2, 3)).reshape(-1)
        assert numpy.abs(cur - ref).max() < 1e-10 * numpy.abs(ref).max()
        units = P.launch_units(chain)
        assert [c for _, c in units] == [2, 2] and [u.in_place_possible for u, _ in units] == [False, True]
        for inplace in (False, True):
            temp, sched = P.buffer_schedule(chain, inplace)
            assert temp and sched[0] == sched[1] and sched[2] == sched[3]
            assert sched[0][0] == (1 if inplace else 0) and sched[0][1] != sched[0][0] and sched[3][1] == 1
            assert sched[2][0] == sched[0][1]
    r0 = N.lib.mifft_pair_split(N.F64, N.INTERLEAVED, 256, 256, 256)
    assert r0 in (32, 64)
    chain = P.build_chain(256, 256, 256, N.F64, interleaved=True)
    assert len(chain) == 4 and chain[1].L == r0 and chain[1].M == 256 // r0 and chain[2].S == 256 * r0
    split = P.build_chain(256, 256, 256, N.F64, interleaved=False)               # split planes: the 64 x 4 split, 16-column tiles
    assert len(split) == 4 and split[1].L == 64 and N.lib.mifft_pair_split(N.F64, N.SPLIT, 256, 256, 256) == 64
    assert N.lib.mifft_pair_split(N.F32, N.INTERLEAVED, 256, 256, 256) == 64 and N.lib.mifft_pair_split(N.F32, N.SPLIT, 256, 256, 256) == 0
    assert N.lib.mifft_pair_split(N.F64, N.INTERLEAVED, 256, 256, 1) == 0 and N.lib.mifft_pair_split(N.F64, N.INTERLEAVED, 100, 256, 256) == 0


def test_round3_entry_points_reject_bad_arguments_without_touching_the_gpu():
    """The pass-pair, tiled N-D and mixed-radix launchers validate before any HIP call: wrong shapes are MIFFT_E_UNSUPPORTED,
    malformed arguments MIFFT_E_INVALID (no GPU needed: this is the host side of the C ABI)."""
    import ctypes
    from pyfft_amd import _native as N

    def row(L, outer):
        p = N.MifftPass()
        p.kind, p.precision, p.layout, p.L, p.M, p.S, p.outer = N.PASS_ROW, N.F64, N.INTERLEAVED, L, 1, 1, outer
        p.outer_stride_in = p.outer_stride_out = L
        p.scale, p.tw_L = 1.0, 16
        return p

    def col(L, M, S, outer, stride):
        p = N.MifftPass()
        p.kind, p.precision, p.layout, p.L, p.M, p.S, p.outer = N.PASS_COL, N.F64, N.INTERLEAVED, L, M, S, outer
        p.outer_stride_in = p.outer_stride_out = stride
        p.scale, p.tw_L, p.tw_lo, p.tw_hi, p.tw_shift = 1.0, 16, 16, 16, 4
        return p

    # the pair of BASELINE config 4 is recognised ...
    x0, y0 = row(256, 256 * 256), col(32, 8, 256, 256, 65536)
    assert N.lib.mifft_pass_pair_supported(ctypes.byref(x0), ctypes.byref(y0)) == 0
    y1, z1 = col(8, 1, 8192, 256, 65536), col(256, 1, 65536, 1, 1 << 24)
    assert N.lib.mifft_pass_pair_supported(ctypes.byref(y1), ctypes.byref(z1)) == 0
    # ... two passes that are not consecutive passes of one plan are not, a shape without kernels is not
    assert N.lib.mifft_pass_pair_supported(ctypes.byref(x0), ctypes.byref(z1)) == N.E_UNSUPPORTED
    bad = col(16, 16, 256, 256, 65536)
    assert N.lib.mifft_pass_pair_supported(ctypes.byref(x0), ctypes.byref(bad)) == N.E_UNSUPPORTED
    assert N.lib.mifft_launch_pass_pair(ctypes.byref(x0), ctypes.byref(bad), 16, None, 32, None, None) == N.E_UNSUPPORTED
    # the (ROW x, COL y) pair never runs in place; buffers must be aligned
    assert N.lib.mifft_launch_pass_pair(ctypes.byref(x0), ctypes.byref(y0), 64, None, 64, None, None) == N.E_INVALID
    assert "in place" in N.last_error()
    assert N.lib.mifft_launch_pass_pair(ctypes.byref(x0), ctypes.byref(y0), 8, None, 64, None, None) == N.E_INVALID
    # tiled N-D: a shape without a tiled kernel, an inconsistent tiling
    nd = N.MifftPass()
    nd.kind, nd.precision, nd.layout, nd.L, nd.M, nd.S, nd.outer = N.PASS_ND, N.F32, N.INTERLEAVED, 16, 16, 1, 32
    nd.scale, nd.tw_L, nd.tw_lo = 1.0, 16, 16
    t = N.MifftTiling()
    t.pitch_y, t.pitch_z, t.parent_elems, t.cx, t.cy, t.cz = 64, 64 * 64, 64 * 64, 4, 4, 1
    assert N.lib.mifft_nd_tiled_supported(N.F32, 16, 16, 1) == 0 and N.lib.mifft_nd_tiled_supported(N.F32, 16, 4, 1) == N.E_UNSUPPORTED
    t.pitch_y = 32                                   # four tiles of 16 do not fit a row of 32
    assert N.lib.mifft_launch_nd_tiled(ctypes.byref(nd), ctypes.byref(t), 16, 16, None) == N.E_INVALID
    nd.M = 4
    t.pitch_y = 64
    assert N.lib.mifft_launch_nd_tiled(ctypes.byref(nd), ctypes.byref(t), 16, 16, None) == N.E_UNSUPPORTED
    # mixed-radix rows: smooth lengths only, up to 4096 (fp32) / 2048 (fp64)
    assert [N.lib.mifft_mixed_supported(N.F32, n) == 0 for n in (1000, 4096, 4097, 11, 1, 3 * 5 * 7 * 16)] == [True, True, False, False, False, True]
    assert N.lib.mifft_mixed_supported(N.F64, 4096) == N.E_UNSUPPORTED and N.lib.mifft_mixed_supported(N.F64, 2000) == 0
    assert N.lib.mifft_launch_mixed_rows(N.F32, 1023, 4, 1023, 1023, 16, 16, 16, 0, 1.0, None) == N.E_UNSUPPORTED
    assert N.lib.mifft_launch_mixed_rows(N.F32, 1000, 4, 999, 1000, 16, 16, 16, 0, 1.0, None) == N.E_INVALID     # stride < n
    assert N.lib.mifft_launch_mixed_lines(N.F32, 1000, 4, 0, 16, 16, 16, 0, 0, 1.0, None) == N.E_INVALID
    # long smooth lengths: a split n1 * n2 whose factors both fit a tile with enough adjacent lines, or unsupported
    n1, n2 = ctypes.c_int32(0), ctypes.c_int32(0)
    for prec, n in ((N.F32, 30000), (N.F32, 5000), (N.F64, 30000), (N.F32, 196608), (N.F64, 3000)):
        assert N.lib.mifft_mixed_long_split(prec, n, ctypes.byref(n1), ctypes.byref(n2)) == 0
        assert n1.value * n2.value == n
        assert N.lib.mifft_mixed_supported(prec, n1.value) == 0 and N.lib.mifft_mixed_supported(prec, n2.value) == 0
    for n in (12289, 4098, 1000000, 3, 1 << 25):              # a large prime factor; no split with 8 lines per tile; tiny; beyond 2^24
        assert N.lib.mifft_mixed_long_split(N.F32, n, ctypes.byref(n1), ctypes.byref(n2)) == N.E_UNSUPPORTED
    assert N.lib.mifft_launch_mixed_long(N.F32, 120, 250, 4, 16, 16, 32, 16, 16, 16, 16, 8, 0, 1.0, None) == N.E_INVALID   # mid == in
    assert N.lib.mifft_launch_mixed_long(N.F32, 121, 250, 4, 16, 32, 32, 16, 16, 16, 16, 8, 0, 1.0, None) == N.E_UNSUPPORTED
    assert N.lib.mifft_launch_mixed_long(N.F32, 120, 250, 4, 16, 32, 32, 16, 16, 16, None, 8, 0, 1.0, None) == N.E_INVALID
    # Bluestein in one launch: a smooth padded length >= 2 n - 1 inside one tile, or unsupported
    m = ctypes.c_int32(0)
    for prec, n in ((N.F32, 1009), (N.F32, 17), (N.F32, 2048), (N.F64, 1023), (N.F32, 2)):
        assert N.lib.mifft_bluestein_padded(prec, n, ctypes.byref(m)) == 0
        assert m.value >= 2 * n - 1 and N.lib.mifft_mixed_supported(prec, m.value) == 0
    # round 4: padded rows up to 10000 (fp32) / 5000 (fp64) points -- one row per work-group in up to 160 KB of LDS
    for prec, n in ((N.F32, 2049), (N.F32, 4099), (N.F32, 5000), (N.F64, 1025), (N.F64, 2500)):
        assert N.lib.mifft_bluestein_padded(prec, n, ctypes.byref(m)) == 0 and m.value >= 2 * n - 1
        assert m.value <= (10000 if prec == N.F32 else 5000)
    assert N.lib.mifft_bluestein_padded(N.F32, 5001, ctypes.byref(m)) == N.E_UNSUPPORTED
    assert N.lib.mifft_bluestein_padded(N.F64, 2501, ctypes.byref(m)) == N.E_UNSUPPORTED
    # smooth N-D shapes in one launch: every axis smooth, at least two axes, the transform inside one tile
    assert [N.lib.mifft_mixed_nd_supported(N.F32, *xyz) == 0 for xyz in ((100, 100, 1), (60, 60, 1), (30, 20, 10), (100, 101, 1), (1000, 1, 1),
                                                                            (128, 100, 1), (160, 128, 1), (60, 60, 60))] == [True, True, True, False, False, True, False, False]
    assert N.lib.mifft_mixed_nd_supported(N.F64, 100, 100, 1) == N.E_UNSUPPORTED and N.lib.mifft_mixed_nd_supported(N.F64, 70, 70, 1) == 0
    assert N.lib.mifft_launch_mixed_nd(N.F32, 100, 100, 1, 4, 16, 16, None, 16, None, 0, 1.0, None) == N.E_INVALID        # x table missing
    assert N.lib.mifft_launch_mixed_nd(N.F32, 100, 101, 1, 4, 16, 16, 16, 16, None, 0, 1.0, None) == N.E_UNSUPPORTED
    assert N.lib.mifft_launch_bluestein_rows(N.F32, 1009, 2000, 4, 1009, 1009, 16, 16, 16, 16, 16, 0, 1.0, None) == N.E_UNSUPPORTED  # m < 2n-1
    assert N.lib.mifft_launch_bluestein_rows(N.F32, 1009, 2048, 4, 1000, 1009, 16, 16, 16, 16, 16, 0, 1.0, None) == N.E_INVALID      # stride < n
    assert N.lib.mifft_launch_bluestein_rows(N.F32, 1009, 2048, 4, 1009, 1009, 16, 16, 16, None, 16, 0, 1.0, None) == N.E_INVALID
    # the fp64 strided passes of 2048 points exist, the pair split answers per layout
    assert N.lib.mifft_pass_supported(N.PASS_COL, N.F64, 2048, 0) == 0
    assert N.lib.mifft_pair_split(N.F64, N.SPLIT, 256, 256, 256) == 64 and N.lib.mifft_pair_split(N.F32, N.INTERLEAVED, 128, 128, 128) == 32


# ---- round 4: the planner sizes everything from the device's properties (pyfft_amd/machine.py) --------------------------
from helpers import FakeContext as _FakeContext      # (a context without a device; shared with tests/kernel_coverage.py)


def _strategy_on(machine, shape, dtype, batch):
    from pyfft_amd.plan import FFTPlan
    plan = FFTPlan(_FakeContext(machine), shape, dtype=dtype)
    return plan, plan._select_strategy(batch)


def test_planner_constants_come_from_the_device():
    """On the full part (256 CUs, 8 XCDs, 256 MiB last-level cache) the machine model gives back the measured constants of
    rounds 1-3 and the round-4 ring rule; on a partition (32 CUs, one XCD, 32 MiB of cache) every ring, chunk and threshold
    shrinks with the cache, the XCD-cooperative strategies disappear, and nothing is sized for a machine that is not there;
    without a last-level cache every plan is the plain chain (the reference's loop, pyfft/plan.py:217-248)."""
    import numpy
    from pyfft_amd.machine import Machine
    full = Machine(256, 8, 4 << 20, 256 << 20)
    assert (full.ring_bytes, full.pipeline_chunk_bytes, full.slab_bytes, full.chain_max_bytes, full.write_through_max_bytes) == \
        (224 << 20, 64 << 20, 128 << 20, 256 << 20, 128 << 20)
    assert full.xcd_cooperative
    c64, c128 = numpy.complex64, numpy.complex128
    assert _strategy_on(full, (1 << 20,), c64, 4096)[1] == ("fused2", 14, 28, 512)            # BASELINE config 2, as in rounds 2-3
    assert _strategy_on(full, (1 << 19,), c64, 2048)[1] == ("fused2", 28, 56, 512)            # round 4: the ring fills the cache
    assert _strategy_on(full, (1 << 18,), c64, 4096)[1] == ("fused2", 56, 112, 512)           # 32-column tiles: 16 first-pass tiles per transform
    assert _strategy_on(full, (1 << 22,), c64, 256)[1] == ("fused2", 4, 7, 256)               # BASELINE config 5's chunk
    assert _strategy_on(full, (1 << 21,), c64, 512)[1] == ("fused2", 8, 14, 512)              # 16 MiB transforms: the cache caps the ring
    assert _strategy_on(full, (1 << 17,), c64, 8192)[1] == ("fused2", 112, 224, 512)          # 2^16 / 2^17: 8 tiles of 32 columns per pass
    assert _strategy_on(full, (1 << 16,), c64, 16384)[1] == ("fused2", 112, 224, 512)
    assert _strategy_on(full, (1 << 17,), numpy.float32, 8192)[1] == ("fused2", 112, 224, 512)  # split planes: sibling tiles interleaved at lane level, 512-thread work-groups
    assert _strategy_on(full, (1 << 16,), numpy.float32, 8192)[1] == ("fused2", 112, 224, 512)
    assert _strategy_on(full, (1 << 20,), numpy.float32, 512)[1] == ("fused2", 14, 28, 256)    # (L = 1024: one such work-group per CU)
    assert _strategy_on(full, (1 << 18,), numpy.float32, 4096)[1] == ("fused2", 56, 112, 512)
    assert _strategy_on(full, (256, 256), c64, 4096)[1] == ("fused2", 112, 224, 512)          # a 256-point axis on the 2-D persistent kernel
    assert _strategy_on(full, (512, 256), c128, 1024)[1][0] == "fused2"
    assert _strategy_on(full, (1 << 22,), numpy.complex128, 64)[1][0] == "pipelined"          # fp64 2^22: a ring of three 64 MiB slots loses
    assert _strategy_on(full, (1 << 21,), numpy.complex128, 128)[1] == ("fused2", 4, 7, 256)
    assert _strategy_on(full, (1 << 22,), numpy.float64, 64)[1][0] == "pipelined"             # (split planes: no kernel)
    assert _strategy_on(full, (1024, 1024), c64, 512)[1] == ("fused2", 14, 28, 512)           # BASELINE config 3
    assert _strategy_on(full, (128, 128, 128), c64, 64)[1][0] == "fusedp"
    assert _strategy_on(full, (128, 128, 128), c128, 32)[1][0] == "fusedp"
    assert _strategy_on(full, (256, 256, 256), c128, 64)[1][0] == "pipelined"                 # BASELINE config 4 (256 MiB per transform)
    assert _strategy_on(full, (1024, 1024), c64, 4)[1] == ("chain",)                          # 32 MiB: below the chain threshold
    assert _strategy_on(full, (1 << 20,), c64, 32)[1] == ("chain",)                           # exactly the cache size per side

    part = Machine(32, 1, 4 << 20, 32 << 20)
    assert not part.xcd_cooperative and part.ring_bytes == 28 << 20 and part.chain_max_bytes == 32 << 20
    for shape, dtype, batch in (((1 << 20,), c64, 4096), ((1 << 19,), c64, 2048), ((1 << 18,), c64, 4096), ((1 << 17,), c64, 8192),
                                ((1 << 22,), c64, 256), ((1024, 1024), c64, 512), ((128, 128, 128), c64, 64), ((256, 256, 256), c128, 8),
                                ((1 << 20,), c128, 512), ((1024,), c64, 1 << 16)):
        plan, st = _strategy_on(part, shape, dtype, batch)
        item = plan._params.size * plan._params.complex_nbytes
        assert st[0] in ("chain", "pipelined", "fused2", "fusedp"), (shape, st)                # never xcd2 / per-XCD lists
        if st[0] in ("fused2", "fusedp"):
            lag, ring, grid = st[1:4]
            assert ring * item <= part.ring_bytes and 1 <= lag < ring and grid in (32, 64) and batch >= 2 * ring, (shape, st)
        if st[0] == "pipelined":
            assert st[1] * item <= max(item, part.pipeline_chunk_bytes), (shape, st)
    assert _strategy_on(part, (1 << 20,), c64, 4096)[1] == ("pipelined", 1, 2, 0)              # an 8 MiB transform: three ring slots are no ring
    assert _strategy_on(part, (1 << 18,), c64, 4096)[1] == ("fused2", 7, 14, 64)               # 2 MiB transforms: all 14 slots
    assert _strategy_on(part, (1 << 20,), c64, 4)[1] == ("chain",)                             # 32 MiB per side = this device's cache
    assert _strategy_on(part, (1 << 20,), c64, 16)[1] == ("pipelined", 1, 2, 0)                # (the full part runs the chain here)

    bare = Machine(64, 2, 4 << 20, 0)
    for shape, dtype, batch in (((1 << 20,), c64, 4096), ((1024, 1024), c64, 512), ((128, 128, 128), c64, 64), ((1 << 17,), c64, 8192)):
        assert _strategy_on(bare, shape, dtype, batch)[1] == ("chain",)
    # the model's geometry helper: no ring below four slots, lag counted in first-pass tiles per work-group wave
    assert full.fused_geometry(8 << 20, 64, 2) == (14, 28, 512) and full.fused_geometry(64 << 20, 128, 1) is None
    assert full.fused_geometry(32 << 20, 128, 1) == (4, 7, 256)


def test_round4_entry_points_reject_bad_arguments_without_touching_the_gpu():
    """mifft_fused_sync validation and the fused pass-pair launcher (host side of the C ABI; every call returns before HIP)."""
    import ctypes
    from pyfft_amd import _native as N

    def col(L, M, S, outer, stride, prec=N.F32):
        p = N.MifftPass()
        p.kind, p.precision, p.layout, p.L, p.M, p.S, p.outer = N.PASS_COL, prec, N.INTERLEAVED, L, M, S, outer
        p.outer_stride_in = p.outer_stride_out = stride
        p.scale, p.tw_L, p.tw_lo, p.tw_hi, p.tw_shift = 1.0, 16, 16, 16, 10
        return p

    p0, p1 = col(1024, 1024, 1, 64, 1 << 20), col(1024, 1, 1024, 64, 1 << 20)
    ok = N.MifftFusedSync(4096, 8192, 12288)
    byref = ctypes.byref
    # null / misaligned / aliased counter sets, bad lag / ring combinations
    for sync, what in ((N.MifftFusedSync(None, None, None), "null counters"), (N.MifftFusedSync(4096 + 64, None, None), "256-byte"),
                       (N.MifftFusedSync(4096, 4096, None), "second buffer"), (N.MifftFusedSync(4096, 8192, 4098), "error word")):
        assert N.lib.mifft_launch_fused2(byref(p0), byref(p1), 16, None, 32, None, 64, None, 28, 14, byref(sync), 512, None) == N.E_INVALID
        assert what in N.last_error(), N.last_error()
    assert N.lib.mifft_launch_fused2(byref(p0), byref(p1), 16, None, 32, None, 64, None, 28, 14, None, 512, None) == N.E_INVALID
    assert N.lib.mifft_launch_fused2(byref(p0), byref(p1), 16, None, 32, None, 64, None, 14, 14, byref(ok), 512, None) == N.E_INVALID   # lag == ring
    # two alternating counter sets need an error word of their own (the next launch zeroes the default one, word 1 of line 0)
    assert N.lib.mifft_launch_fused2(byref(p0), byref(p1), 16, None, 32, None, 64, None, 28, 14, byref(N.MifftFusedSync(4096, 8192, None)), 512, None) == N.E_INVALID
    assert "error word of their own" in N.last_error(), N.last_error()
    # the development forms (sequential list, per-XCD lists, XCD-resident kernel) exist in `make DEV=1` builds only; the default
    # build says so loudly and mifft_has_feature tells beforehand
    dev = [N.lib.mifft_has_feature(f) for f in (N.FEATURE_XCD2, N.FEATURE_FUSED2X, N.FEATURE_SEQUENTIAL_LIST)]
    assert dev in ([0, 0, 0], [1, 1, 1]) and N.lib.mifft_has_feature(99) == 0
    seq = N.lib.mifft_launch_fused2(byref(p0), byref(p1), 16, None, 32, None, 64, None, 28, 0, byref(ok), 512, None)    # sequential list: ring == outer
    assert seq == (N.E_INVALID if dev[2] else N.E_UNSUPPORTED)
    for sync in (None, byref(N.MifftFusedSync(4096, 4096, None))):
        rc = N.lib.mifft_launch_fused2x(byref(p0), byref(p1), 16, None, 32, None, 64, 8, 4, sync, 512, None)
        assert rc == (N.E_INVALID if dev[1] else N.E_UNSUPPORTED)
    if not dev[0]:
        assert N.lib.mifft_launch_xcd2(byref(p0), byref(p1), 16, None, 32, None, 256, 256, 1, None) == N.E_UNSUPPORTED
        assert "DEV=1" in N.last_error()
    # the persistent pass-pair form exists for the shapes with every axis in {64, 128}, both precisions, both layouts
    I, S = N.INTERLEAVED, N.SPLIT
    assert N.lib.mifft_fused_pair_supported(N.F32, I, 128, 128, 128) == 0 and N.lib.mifft_fused_pair_supported(N.F64, I, 128, 128, 128) == 0
    assert N.lib.mifft_fused_pair_supported(N.F32, I, 256, 256, 256) == N.E_UNSUPPORTED
    assert N.lib.mifft_fused_pair_supported(7, I, 128, 128, 128) == N.E_UNSUPPORTED and N.lib.mifft_fused_pair_supported(N.F32, 5, 128, 128, 128) == N.E_UNSUPPORTED
    for x in (64, 128):
        for y in (64, 128):
            for z in (64, 128):
                want = {64: 16, 128: 32}[y]
                assert N.lib.mifft_fused_pair_split(N.F32, I, x, y, z) == want and N.lib.mifft_fused_pair_split(N.F64, I, x, y, z) == want
                assert N.lib.mifft_fused_pair_split(N.F64, S, x, y, z) == want
                assert N.lib.mifft_fused_pair_split(N.F32, S, x, y, z) == want
    assert N.lib.mifft_fused_pair_split(N.F32, I, 64, 32, 64) == 0 and N.lib.mifft_fused_pair_split(N.F32, I, 32, 32, 32) == 0
    # (round 6: and for (32, 32, 128), interleaved, y = 8 x 4)
    assert N.lib.mifft_fused_pair_split(N.F32, I, 128, 32, 32) == 8 and N.lib.mifft_fused_pair_split(N.F64, I, 128, 32, 32) == 8
    assert N.lib.mifft_fused_pair_split(N.F32, S, 128, 32, 32) == 0
    from pyfft_amd import passes as P
    chain = P.build_chain(128, 128, 128, N.F32, interleaved=True)
    assert [k.pair_with_next for k in chain] == [True, False, True, False]
    descs = (N.MifftPass * 4)()
    for d, k in zip(descs, chain):
        d.kind, d.precision, d.layout, d.L, d.M, d.S = k.kind, N.F32, N.INTERLEAVED, k.L, k.M, k.S
        d.outer, d.outer_stride_in, d.outer_stride_out, d.scale = k.outer_per_batch * 20, k.outer_stride, k.outer_stride, 1.0
        d.tw_L, d.tw_lo, d.tw_hi, d.tw_shift = 16, 16, 16, 4
    assert N.lib.mifft_launch_fused_pair(descs, 16, None, 32, None, 32, 4, 2, byref(ok), 512, None) == N.E_INVALID and "ring" in N.last_error()
    assert N.lib.mifft_launch_fused_pair(descs, 16, None, 32, None, 64, 4, 4, byref(ok), 512, None) == N.E_INVALID
    assert N.lib.mifft_launch_fused_pair(descs, 16, None, 32, None, 64, 4, 2, None, 512, None) == N.E_INVALID
    descs[1].L, descs[1].M = 64, 2                      # not the split the library's kernels use
    assert N.lib.mifft_launch_fused_pair(descs, 16, None, 32, None, 64, 4, 2, byref(ok), 512, None) in (N.E_INVALID, N.E_UNSUPPORTED)
    # device properties carry the memory-system fields the planner reads
    assert {"llc_bytes", "num_xcc"} <= {f[0] for f in N.MifftDeviceProps._fields_}


def test_pyfft_import_name():
    """importable as `pyfft`, VERSION kept (pyfft/__init__.py:1 of the reference); the backend module is pyfft.hip,
    and there is no pyfft.cuda compatibility alias."""
    import importlib
    import pyfft
    assert pyfft.VERSION == (0, 3, 9) and all(isinstance(v, int) for v in pyfft.VERSION)
    from pyfft.hip import Plan
    import pyfft_amd.hip
    assert Plan is pyfft_amd.hip.Plan
    try:
        importlib.import_module("pyfft.cuda")
    except ImportError:
        pass
    else:
        raise AssertionError("pyfft.cuda must not exist")


def test_round5_debug_switches_are_per_thread_over_a_process_default():
    """mifft_debug_set changes a development switch for the calling thread only; a thread that never set the key sees the process
    default (mifft_debug_set_default) -- a measurement in one thread cannot change the kernels of another thread's plan."""
    import threading
    from pyfft_amd import _native as N
    key = N.DEBUG_ALT_ROWS
    assert N.lib.mifft_debug_get(key) == 0
    seen = []

    def other(tag):
        seen.append((tag, N.lib.mifft_debug_get(key)))

    def run(tag):
        t = threading.Thread(target=other, args=(tag,))
        t.start()
        t.join()

    try:
        assert N.lib.mifft_debug_set(key, 2) == 0
        run("after the main thread's set")
        assert N.lib.mifft_debug_get(key) == 2
        assert N.lib.mifft_debug_set_default(key, 3) == 0
        run("after the default changed")
        assert N.lib.mifft_debug_get(key) == 2            # the thread's own value wins
    finally:
        N.lib.mifft_debug_set_default(key, 0)
        N.lib.mifft_debug_set(key, 0)
    assert seen == [("after the main thread's set", 0), ("after the default changed", 3)]
    assert N.lib.mifft_debug_set(N.DEBUG_PREFETCH + 1, 1) == N.E_INVALID and N.lib.mifft_debug_set_default(-1, 1) == N.E_INVALID


def test_round5_capture_entry_points_reject_bad_arguments_without_touching_the_gpu():
    """Stream capture / graph replay shims (include/mifft.h): argument errors are negative library codes, never a crash."""
    import ctypes
    from pyfft_amd import _native as N
    assert N.lib.mifft_stream_is_capturing(None, None) == N.E_INVALID
    assert N.lib.mifft_stream_begin_capture(None) == N.E_INVALID and "default stream" in N.last_error()
    assert N.lib.mifft_stream_end_capture(None, None) == N.E_INVALID
    assert N.lib.mifft_graph_launch(None, None) == N.E_INVALID
    assert N.lib.mifft_graph_destroy(None) == 0
    assert N.ABI_VERSION == 5 and N.lib.mifft_abi_version() == 5


def test_round5_no_split_rowfirst_follows_the_environment_into_the_library(monkeypatch):
    """PYFFT_AMD_NO_SPLIT_ROWFIRST decides the planner's tile count AND the launcher's kernel: the planner's query forwards it to
    the calling thread's native switch (the two used to disagree)."""
    from pyfft_amd import _debug as D
    from pyfft_amd import _native as N
    monkeypatch.delenv("PYFFT_AMD_NO_SPLIT_ROWFIRST", raising=False)
    assert D.no_split_rowfirst() is False and N.lib.mifft_debug_get(N.DEBUG_NO_ROWFIRST) == 0
    monkeypatch.setenv("PYFFT_AMD_NO_SPLIT_ROWFIRST", "1")
    assert D.no_split_rowfirst() is True and N.lib.mifft_debug_get(N.DEBUG_NO_ROWFIRST) == 1
    monkeypatch.delenv("PYFFT_AMD_NO_SPLIT_ROWFIRST")
    assert D.no_split_rowfirst() is False and N.lib.mifft_debug_get(N.DEBUG_NO_ROWFIRST) == 0


def test_strategy_snapshot_of_the_table_driven_planner():
    """FFTPlan._select_strategy is a lookup in pyfft_amd/tuning_gfx950.json since round 5.  It must answer what the round-4 planner
    (nested literals) answered for every shape of profiles/r04_long_1d_sizes.log / r04_second_batch_shapes.log / r04_t_tail_survey.log
    at seven buffer sizes on three devices, and on the full part under the development switches the choice depends on: tests/golden/strategy_snapshot.json.gz, written by make_strategy_snapshot.py BEFORE the move."""
    import json
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_strategy_snapshot as snap
    import gzip
    from pyfft_amd import _native as N
    with gzip.open(os.path.join(ROOT, "tests", "golden", "strategy_snapshot.json.gz"), "rt") as f:
        want = json.load(f)
    got = snap.snapshot(with_chain_class=True)
    assert len(got) == len(want) > 30000
    # (the per-XCD work lists are part of `make DEV=1` builds only: 18 rows of the 16-column A/B mode)
    lists = N.lib.mifft_has_feature(N.FEATURE_FUSED2X) == 1
    # shapes whose CHAIN changed in round 5 (one pass pair instead of a third launch: (4096, 256), (32, 32, 2048) ...) no longer run
    # their leading passes slab-wise; they stay on the plain chain / the pipelined chunks
    new_chain = [g for g in got if g[6]]
    # (2.6 k of the 38 k rows by the end of round 5: the 4096-point y axes, short (y, z) behind long rows, the 3-D shapes with a 256-point y axis)
    assert 0 < len(new_chain) < 4000 and all(g[5][0] in ("chain", "pipelined") for g in new_chain)
    # round 6: the one shape of the grid beyond a pipeline chunk with a ROW x + COL y + COL z chain runs its planes on the persistent 2-D
    # kernels (strategy "fused2z") where the round-4 planner answered chain / pipelined slabs
    planes = [g for g in got if g[5][0] == "fused2z"]
    assert 0 < len(planes) < 200 and all(tuple(g[2]) == (512, 512, 512) and w[5][0] in ("chain", "pipelined") for w, g in zip(want, got) if g[5][0] == "fused2z")
    # ... and (32, 32, 128), a shape of the reference's benchmark list, got its persistent two-pair kernel (round-4 answer: the pipelined chunks)
    cube32 = [g for w, g in zip(want, got) if tuple(g[2]) == (32, 32, 128) and g[5][0] == "fusedp"]
    assert cube32 and all(w[5][0] in ("chain", "pipelined") for w, g in zip(want, got) if tuple(g[2]) == (32, 32, 128) and g[5][0] == "fusedp")
    bad = [(w, g) for w, g in zip(want, got) if not g[6] and g[5][0] != "fused2z" and not (tuple(g[2]) == (32, 32, 128) and g[5][0] == "fusedp") and
           w != json.loads(json.dumps(g[:6])) and (lists or w[5][0] != "fused2x")]
    assert not bad, bad[:10]
    assert not lists or sum(1 for g in got if g[5][0] == "fused2x") == 18
    assert set(r[5][0] for r in want) == {"chain", "pipelined", "fused2", "fusedp", "fused2x"}


def test_planner_follows_a_doctored_tuning_table():
    """The planner holds no measured literal of its own: a table with other fractions / rules (what tools/fused_sweep.py --emit
    writes after a sweep on another part) changes the strategies accordingly, and a broken table is refused when it is loaded."""
    import copy
    import numpy
    from pyfft_amd import tuning
    from pyfft_amd.machine import Machine
    from pyfft_amd.plan import FFTPlan
    base = tuning.default()
    assert base.source.endswith("tuning_gfx950.json") and len(base.rules) >= 20 and all("evidence" in r for r in base.rules)
    c64 = numpy.complex64

    def on(table, shape, dtype, batch):
        mach = Machine(256, 8, 4 << 20, 256 << 20, tuning=tuning.Tuning(table, "doctored"))
        return FFTPlan(_FakeContext(mach), shape, dtype=dtype)._select_strategy(batch)

    assert on(base.table, (1 << 20,), c64, 4096) == ("fused2", 14, 28, 512)
    t = copy.deepcopy(base.table)
    t["cache_fractions"]["ring"] = [1, 2]                       # half the cache for the ring: 16 slots of 8 MiB -> the cap rule (14 slots)
    t["cache_fractions"]["chain_max"] = [1, 8]                  # chain only up to 32 MiB per side
    t["cache_fractions"]["pipeline_chunk"] = [1, 16]            # 16 MiB chunks
    assert on(t, (1 << 20,), c64, 4096) == ("fused2", 8, 14, 512)
    assert on(t, (1 << 20,), c64, 8) == ("pipelined", 2, 2, 0) and on(base.table, (1 << 20,), c64, 8) == ("chain",)
    t = copy.deepcopy(base.table)
    for r in t["rules"]:
        if r["kind"] == "1d" and r.get("precision") == "f32" and 1024 in r.get("L0", []) and r["cols0"] == 16:
            r["on_request"] = True                              # "the persistent form lost on this part"
    assert on(t, (1 << 20,), c64, 4096)[0] == "pipelined" and on(t, (1 << 18,), c64, 4096)[0] == "fused2"
    t = copy.deepcopy(base.table)
    t["ring_rule"]["lag_waves"] = [7, 2]                        # producers 3.5 waves ahead
    assert on(base.table, (1 << 19,), c64, 2048) == ("fused2", 28, 56, 512)
    assert on(t, (1 << 19,), c64, 2048) == ("fused2", 8, 14, 512)      # the tile rule would ask for 56 / 112, the cache holds 56: the capped ring
    t = copy.deepcopy(base.table)
    del t["rules"][0]["cols0"]
    try:
        tuning.Tuning(t, "broken")
    except ValueError as e:
        assert "cols0" in str(e)
    else:
        raise AssertionError("a rule without cols0 was accepted")


def test_round5_tiny_nd_shapes_take_the_run_time_shaped_kernel(monkeypatch):
    """One-launch N-D shapes whose fixed instance measured slower than the run-time-shaped kernel (profiles/r05_nd2_table_value.log: two- and
    four-point rows) are marked variant 1 by the plan -- always, or in launches beyond write_through_max per side, as the tuning table's
    "nd_generic" lists say; the lists are data: a table without them, or PYFFT_AMD_NO_ND_GENERIC, gives variant 0 everywhere."""
    import copy
    import numpy
    from pyfft_amd import tuning
    from pyfft_amd.machine import Machine
    from pyfft_amd.plan import FFTPlan
    c64, c128 = numpy.complex64, numpy.complex128

    def variant(shape, dtype, batch, table=None):
        mach = Machine(256, 8, 4 << 20, 256 << 20, tuning=tuning.Tuning(table, "doctored") if table is not None else None)
        return FFTPlan(_FakeContext(mach), shape, dtype=dtype)._descriptors(batch, False, False)[0].variant

    assert variant((16, 2), c64, 3) == 1 and variant((16, 2), c64, 1 << 22) == 1              # "always"
    assert variant((2, 8), c64, 100) == 0 and variant((2, 8), c64, 1 << 22) == 1               # "big": 16 points x 8 bytes x 2^22 = 512 MiB
    assert variant((4, 4), c128, 100) == 0 and variant((4, 4), c128, 1 << 22) == 1
    assert variant((16, 16), c64, 1 << 20) == 0 and variant((16, 16, 16), c64, 1 << 16) == 0   # the published shapes stay on their instances
    assert variant((16, 2), numpy.float32, 1 << 22) == 0                                       # planes: lists of their own (f32_split / f64_split)
    # (round 6: the planes lists are empty in the shipped table -- (32, 32) float32 planes left it for the 16-byte dense kernel of
    # csrc/fft_nd2p.hpp --, so the mechanism is shown on a doctored table)
    planes = copy.deepcopy(tuning.default().table)
    planes["nd_generic"]["f32_split"] = {"always": [[32, 32, 1]], "big": []}
    assert variant((32, 32), numpy.float32, 100, planes) == 1 and variant((32, 32), c64, 100, planes) == 0
    assert variant((32, 32), numpy.float32, 100) == 0
    bare = copy.deepcopy(tuning.default().table)
    del bare["nd_generic"]
    assert variant((16, 2), c64, 1 << 22, bare) == 0
    monkeypatch.setenv("PYFFT_AMD_NO_ND_GENERIC", "1")
    assert variant((16, 2), c64, 1 << 22) == 0

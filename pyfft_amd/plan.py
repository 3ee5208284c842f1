"""FFTPlan: plan construction and execution (host side of the hot path).

Drop-in for the reference's pyfft/plan.py: same constructor arguments, same `execute`
signatures bound by data layout, same normalize/scale rule, same buffer contract and sync
policy.  Differences that are not observable through the API: kernels are precompiled HIP
(no run-time code generation), the axis factorisation is the gfx950 one (passes.py), and the
whole pass list is enqueued by one native call (mifft_launch_chain) instead of a Python loop.
"""

import ctypes

import numpy

from . import _debug as D
from . import _native as N
from . import passes as P

_FFT_1D, _FFT_2D, _FFT_3D = 1, 2, 3


def on_plan_device(method):
    """Run a plan method with the plan's device current (Plan(context=i) for a device that is not the caller's current
    one, cuda.py:121-128); a no-op wrapper for plans built on the current device."""
    def wrapped(self, *args, **kwds):
        ctx = self._context
        if not ctx._guard:
            return method(self, *args, **kwds)
        prev = ctx.activate()
        try:
            return method(self, *args, **kwds)
        finally:
            ctx.restore(prev)
    wrapped.__name__ = method.__name__
    wrapped.__doc__ = method.__doc__
    return wrapped


class _FFTParams(object):
    """Plan parameters derived from shape and dtype (plan.py:10-63)."""

    def __init__(self, shape, dtype, context):
        # (the reference's fast_math parameter has no counterpart here -- twiddles come from float64-evaluated tables, hip.Plan's
        # docstring -- so the plan accepts the keyword for signature parity and keeps no state for it)
        self.x, self.y, self.z = shape
        for v in shape:
            if not isinstance(v, (int, numpy.integer)) or isinstance(v, bool) or v < 1:
                raise ValueError("Wrong shape")
        self.size = int(self.x) * int(self.y) * int(self.z)
        self.context = context

        # the reference only checks the product (plan.py:23-24); every axis is checked here
        if not all(P.is_pow2(int(v)) for v in shape):
            raise ValueError("Array dimensions must be powers of two")
        if self.size < 2:
            raise ValueError("Array must have at least two elements")

        try:
            dt = numpy.dtype(dtype)
        except TypeError:
            raise ValueError("Data type " + str(dtype) + " is not supported")
        if dt == numpy.complex64 or dt == numpy.float32:      # plan.py:26-32
            self.split = (dt == numpy.float32)
            self.precision = N.F32
            self.scalar_dtype = numpy.dtype(numpy.float32)
            self.complex_dtype = numpy.dtype(numpy.complex64)
        elif dt == numpy.complex128 or dt == numpy.float64:  # plan.py:33-46
            self.split = (dt == numpy.float64)
            self.precision = N.F64
            self.scalar_dtype = numpy.dtype(numpy.float64)
            self.complex_dtype = numpy.dtype(numpy.complex128)
        else:
            raise ValueError("Data type " + str(dtype) + " is not supported")
        self.scalar_nbytes = self.scalar_dtype.itemsize
        self.complex_nbytes = self.complex_dtype.itemsize
        self.layout = N.SPLIT if self.split else N.INTERLEAVED


def _twiddle_table(n, count, step, complex_dtype):
    """w(n)^(k*step), k < count, evaluated in float64 then rounded to the working precision.
    (The reference evaluates sincos on the device in working precision, kernel.mako:36-44.)"""
    k = (numpy.arange(count, dtype=numpy.int64) * step) % n
    # reduce to the first octant-free form: exact integer phase, float64 trig
    ang = -2.0 * numpy.pi * (k.astype(numpy.float64) / float(n))
    return (numpy.cos(ang) + 1j * numpy.sin(ang)).astype(complex_dtype)


def normalize_shape(shape):
    """Shape normalisation: x is the fastest-varying (last numpy) axis (plan.py:73-89)."""
    if isinstance(shape, (int, numpy.integer)) and not isinstance(shape, bool):
        return _FFT_1D, (int(shape), 1, 1)
    if isinstance(shape, tuple):
        if len(shape) == 1:
            return _FFT_1D, (shape[0], 1, 1)
        if len(shape) == 2:
            return _FFT_2D, (shape[1], shape[0], 1)
        if len(shape) == 3:
            return _FFT_3D, tuple(reversed(shape))
    raise ValueError("Wrong shape")


class FFTPlan(object):
    """Class for FFT plan preparation and execution (plan.py:66-284)."""

    @staticmethod
    def validate(shape, dtype=numpy.complex64, normalize=True, wait_for_finish=None, fast_math=True, scale=1.0):
        """Raise the reference's ValueErrors (plan.py:24,48,87,89) without touching a device."""
        _, xyz = normalize_shape(shape)
        _FFTParams(xyz, dtype, None)

    def __init__(self, context, shape, dtype=numpy.complex64, normalize=True,
                 wait_for_finish=None, fast_math=True, scale=1.0):
        self._dim, shape = normalize_shape(shape)

        self._context = context
        self._params = _FFTParams(shape, dtype, context)
        self._normalize = normalize
        self._scale = float(scale)
        self._wait_for_finish = wait_for_finish

        self._tempmemobj = None      # always one interleaved buffer (see _prepare)
        self._last_batch_size = 0
        self._desc_cache = {}
        self._strategy = ("chain",)
        self._counters = None        # two alternating counter sets of the persistent launches (mifft_fused_sync)
        self._counter_bytes = 0
        self._counter_set = 0
        self._counters_clean = False
        self._errword = None         # pinned host word the persistent kernels report a dependency time-out in
        self._xcd2_scratch = None
        self._xcd2_disabled = False
        self._mailbox = None
        self._last_call_key = None
        self._last_call = None
        self._side_streams = None
        self._side_events = None
        self._scratch_ready = False  # scratch of the current batch's strategy exists (allocated by the first execute that needs it)
        self._captured = False       # some execute() of this plan was recorded into a graph (hip.Graph, torch.cuda.graph())
        self._capture_keepalive = [] # scratch a captured launch baked the addresses of: kept until the plan itself goes away

        if self._params.split:
            self.execute = self._executeSplit
        else:
            self.execute = self._executeInterleaved

        on_plan_device(FFTPlan._generateKernelCode)(self)

    # ------------------------------------------------------------------------------------
    def _generateKernelCode(self):
        """Select the pass chain and upload twiddle tables (plan.py:111-133; nothing is
        compiled here -- the kernels are ahead-of-time HIP)."""
        p = self._params
        self._kernels = P.build_chain(int(p.x), int(p.y), int(p.z), p.precision, interleaved=not p.split)
        self._paired = any(k.pair_with_next for k in self._kernels)
        self._temp_buffer_needed = any(not u.in_place_possible for u, _ in P.launch_units(self._kernels))
        # fp32 split planes: 16 columns of a plane are 64-byte segments, so multi-pass plans detour through an
        # interleaved temp even when every pass could run in place (passes.buffer_schedule)
        self._via_temp = (p.split and p.precision == N.F32 and len(self._kernels) >= 2 and not self._temp_buffer_needed)
        if self._via_temp:
            self._temp_buffer_needed = True
        # 3-D transforms larger than the Infinity Cache: the leading x / y passes work plane by plane, so they can run
        # slab by slab (a few z planes) through the pipelined launcher before the z passes run over whole transforms --
        # the x -> y intermediate then stays on die.  (_slab_passes leading passes, every pass in place capable.)
        self._slab_passes = 0
        if int(p.z) > 1 and not self._temp_buffer_needed and not self._paired and \
                p.size * p.complex_nbytes > max(1, self._context.machine.pipeline_chunk_bytes):
            k = 0
            while k < len(self._kernels) and self._kernels[k].kind != N.PASS_ND and \
                    self._kernels[k].axis in (P.X_DIRECTION, P.Y_DIRECTION):
                k += 1
            if 2 <= k < len(self._kernels):
                self._slab_passes = k

        self._tables = {}      # key -> device allocation
        self._table_ptrs = self._pass_tables(self._kernels)   # per pass: (tw_L, tw_lo, tw_hi, shift)
        # Round 5: shapes of four two-per-CU tiles (65536 points fp32: (256, 256), (32, 32, 64) ...) have a ONE-launch kernel that runs
        # four work-groups per transform (csrc/fft_nd2z.hpp) -- for interleaved data, out of place only (a work-group overwrites
        # planes its partners still read).  The plan keeps its chain for in-place executes and this one-pass list for the others.
        self._oop_nd = None
        self._oop_tables = None
        self._oop_any_size = False
        # (round 6: split-complex planes likewise, csrc/fft_nd2zp.hpp)
        v_oop, v_any = (N.VARIANT_SPLIT_OUT_OF_PLACE, N.VARIANT_SPLIT_OUT_OF_PLACE_ANY_SIZE) if p.split else \
            (N.VARIANT_OUT_OF_PLACE_ONLY, N.VARIANT_OUT_OF_PLACE_ANY_SIZE)
        if len(P.launch_units(self._kernels)) >= 2 and \
                N.lib.mifft_nd_shape_supported(p.precision, int(p.x), int(p.y), int(p.z), v_oop) == 0:
            self._oop_nd = [P.PassSpec(N.PASS_ND, P.X_DIRECTION, p.size, int(p.x), int(p.y), int(p.z), 1, p.size, True)]
            self._oop_tables = self._pass_tables(self._oop_nd)
            # (two halves on two-per-CU tiles: better than the two launches at every size; four quarters: beyond half the cache per side)
            self._oop_any_size = N.lib.mifft_nd_shape_supported(p.precision, int(p.x), int(p.y), int(p.z), v_any) == 0
        # 3-D shapes whose chain is a plane pass + a strided z pass but that have a persistent two-pair kernel (64- and 128-point axes,
        # csrc/fft_fusedp2.hip): the four-pass list with the y axis factored R0 x R1 exists for that launch alone
        self._pair_alt = None
        self._pair_alt_tables = None
        if not self._paired and min(int(p.x), int(p.y), int(p.z)) > 1 and not D.no_fusedp_alt():
            r0 = N.lib.mifft_fused_pair_split(p.precision, p.layout, int(p.x), int(p.y), int(p.z))
            if r0 > 0:
                self._pair_alt = P.pair_chain(int(p.x), int(p.y), int(p.z), r0)
                self._pair_alt_tables = self._pass_tables(self._pair_alt)

    def _pass_tables(self, kernels):
        p = self._params
        ptrs = []
        for k in kernels:
            if k.kind == N.PASS_ND:
                # one w(len)^k table per axis (x, y, z) in the tw_L / tw_lo / tw_hi slots
                tabs = [self._device_table(("L", n), lambda L=n: _twiddle_table(L, L, 1, p.complex_dtype)) if n > 1 else None
                        for n in (k.L, k.M, k.S)]
                ptrs.append((tabs[0], tabs[1], tabs[2], 0))
                continue
            twL = self._device_table(("L", k.L), lambda L=k.L: _twiddle_table(L, L, 1, p.complex_dtype))
            if k.M > 1:
                n = k.curr_n
                shift = (P.log2(n) + 1) // 2
                lo = self._device_table(("lo", n, shift),
                                        lambda n=n, s=shift: _twiddle_table(n, 1 << s, 1, p.complex_dtype))
                hi = self._device_table(("hi", n, shift),
                                        lambda n=n, s=shift: _twiddle_table(n, n >> s, 1 << s, p.complex_dtype))
                ptrs.append((twL, lo, hi, shift))
            else:
                ptrs.append((twL, None, None, 0))
        return ptrs

    def _device_table(self, key, make):
        if key not in self._tables:
            host = numpy.ascontiguousarray(make())
            mem = self._context.allocate_raw(host.nbytes)
            self._context.upload(mem, host)
            self._tables[key] = mem
        return self._context.pointer_of(self._tables[key])

    # ------------------------------------------------------------------------------------
    def _scale_factor(self, inverse):
        """Multiplier applied by the plan's last pass: the reciprocal of the divisor of
        _FFTKernel.getScaleCoeffFunc (kernel.py:23-37)."""
        if not inverse:
            return self._scale
        coeff = (self._params.size if self._normalize else 1.0) * self._scale
        return 1.0 / coeff

    def _descriptors(self, batch, is_inplace, inverse, alt=False):
        """Pass descriptor array for (batch, schedule, direction); cached like the reference's
        kernel.prepare(batch) (kernel.py:85-93).  alt: 1 = the four-pass list of the persistent two-pair launch (_pair_alt), 2 = the
        one-pass list of out-of-place executes (_oop_nd)."""
        key = (batch, is_inplace, inverse, alt)
        d = self._desc_cache.get(key)
        if d is not None:
            return d
        p = self._params
        mach = self._context.machine
        kernels, tables = ((self._kernels, self._table_ptrs), (self._pair_alt, self._pair_alt_tables), (self._oop_nd, self._oop_tables))[int(alt)]
        _, sched = P.buffer_schedule(kernels, is_inplace, self._via_temp and not alt)
        arr = (N.MifftPass * max(1, len(kernels)))()
        last = len(kernels) - 1
        for i, (k, (src, dst), (twL, lo, hi, shift)) in enumerate(zip(kernels, sched, tables)):
            d = arr[i]
            d.kind = k.kind
            d.precision = p.precision
            d.layout = p.layout
            d.inverse = 1 if inverse else 0
            d.L = k.L
            d.variant = 0
            d.M = k.M
            d.S = k.S
            d.outer = k.outer_per_batch * batch
            d.outer_stride_in = k.outer_stride
            d.outer_stride_out = k.outer_stride
            d.scale = self._scale_factor(inverse) if i == last else 1.0
            d.tw_L = twL
            d.tw_lo = lo
            d.tw_hi = hi
            d.tw_shift = shift
            # in-place call: data_out aliases data_in, the schedule only uses indices 1 and 2
            d.src = src
            d.dst = dst
            # the plan-owned temp buffer is always interleaved, also for split-plane plans (include/mifft.h)
            d.flags = N.FLAG_PAIR_WITH_NEXT if k.pair_with_next else 0
            if p.split:
                if src == 2:
                    d.flags |= N.FLAG_SRC_INTERLEAVED
                if dst == 2:
                    d.flags |= N.FLAG_DST_INTERLEAVED
            # multi-pass plans: the first pass reads the input once, nobody re-reads what the last pass writes.  Only
            # while a transform's intermediate can stay in the 256 MiB Infinity Cache (256^3 fp64: 256 MiB per transform,
            # measured 1 % slower with the hints)
            if last >= 1 and (p.size * p.complex_nbytes <= mach.stream_hint_item_bytes or self._slab_passes or self._paired or alt) and not D.no_stream_hints():
                if i == 0:
                    d.flags |= N.FLAG_STREAM_SRC
                if i == last:
                    d.flags |= N.FLAG_STREAM_DST
            # one-launch N-D plans on buffers beyond the Infinity Cache: non-temporal stores of the result (nobody finds it in a
            # cache anyway): (16, 16, 16) 70.4 -> 72.6 %, (128, 128) 64.8 -> 68.1 %, fp64 (128, 128) 58.6 -> 59.9 % at 1 GiB
            # (profiles/r03_c_store_policy.log; the long ROW kernels measured better with plain stores and keep them)
            if last == 0 and k.kind == N.PASS_ND and batch * p.size * p.complex_nbytes > mach.chain_max_bytes and not D.no_stream_hints():
                d.flags |= N.FLAG_STREAM_DST
            # one-launch N-D shapes whose fixed-shape instance measured slower than the run-time-shaped kernel (tiny transforms with two- or
            # four-point rows: (16, 2) 0.094 against 0.693 of the roofline): variant 1 (tuning table "nd_generic", tools/nd2_value_sweep.py)
            if k.kind == N.PASS_ND and not alt and not D.no_nd_generic() and \
                    mach.tuning.nd_runs_generic(p.precision == N.F64, (int(k.L), int(k.M), int(k.S)),
                                                batch * p.size * p.complex_nbytes > mach.write_through_max_bytes, bool(p.split)):
                d.variant = 1
            # small launches: write-through stores, so that the output does not wait dirty in the L2s for the end-of-kernel
            # write-back (32 MiB launches: (16, 16, 16) 14.3 -> 9.1 us, (1024,) 13.5 -> 10.5 us; neutral from 256 MiB on)
            if batch * p.size * p.complex_nbytes <= mach.write_through_max_bytes and not D.no_stream_hints():
                d.flags = (d.flags & ~N.FLAG_STREAM_DST) | N.FLAG_WRITE_THROUGH
        if len(self._desc_cache) > 64:
            self._desc_cache.clear()
        self._desc_cache[key] = arr
        return arr

    # ------------------------------------------------------------------------------------
    # execution strategies (all enqueue the same passes; they differ in how the batch is cut and overlapped)
    #   chain      one launch per pass over the whole batch (the reference's loop, plan.py:217-248)
    #   pipelined  batch cut into cache-sized chunks, chunk i on side stream i % n with its own temp slot
    #              (mifft_launch_chain_pipelined)
    #   fused2     both passes of a long 1-D transform / a big 2-D one in one persistent launch (mifft_launch_fused2)
    #   fused2x    the same with one work list per XCD (mifft_launch_fused2x): on request (2^17 on 16-column tiles: + 2 points)
    #   fusedp     both pass PAIRS of a cache-sized 3-D cube in one persistent launch (mifft_launch_fused_pair)
    #   xcd2       1024 x 1024 fp32: one persistent launch, each transform stays on one XCD between its two HBM
    #              crossings (mifft_launch_xcd2); development only
    # Every size below comes from the device (pyfft_amd/machine.py: fractions of the last-level cache, multiples of the CU count).
    # ---- which persistent launch, if any: the plan's SHAPE CLASS matched against the rules of the tuning table -----------------
    def _shape_class(self):
        """What the rules of pyfft_amd/tuning_gfx950.json are keyed on: "1d" = the two strided passes of one long contiguous axis
        (L0 x L1), "2d" = ROW x + strided COL y of a (ny, nx) plan, "3d" = a plan the library has a persistent two-pair kernel for."""
        p = self._params
        k = self._kernels
        cls = {"precision": "f64" if p.precision == N.F64 else "f32", "layout": "split" if p.split else "interleaved"}
        if self._pair_alt is not None or (self._paired and len(k) == 4 and not p.split and
                                          N.lib.mifft_fused_pair_supported(p.precision, p.layout, int(p.x), int(p.y), int(p.z)) == 0):
            cls.update(kind="3d", planes=int(p.z) * int((self._pair_alt or k)[1].M))          # first-pass items: planes x R1
        elif (len(k) == 2 and int(p.z) == 1 or self._plane_fused()) and k[0].kind == N.PASS_ROW and k[0].L == int(p.x) and \
                k[1].kind == N.PASS_COL and k[1].L == int(p.y) and k[1].M == 1 and k[1].S == int(p.x):
            # planes > 1: the (y, x) planes of a 3-D transform too big for cache-sized chunks -- its ROW x and COL y passes are the
            # two passes of a 2-D plan over batch * nz planes, the z passes follow as plain launches (strategy "fused2z")
            cls.update(kind="2d", ny=int(p.y), nx=int(p.x), M=0, planes=int(p.z))
        elif len(k) == 2 and int(p.y) == 1 and int(p.z) == 1 and k[0].kind == N.PASS_COL and k[1].kind == N.PASS_COL and k[0].S == 1 and \
                k[0].M == k[1].L and k[1].M == 1 and self._temp_buffer_needed:
            cls.update(kind="1d", L0=int(k[0].L), L1=int(k[1].L), M=int(k[0].M))
        else:
            return None
        return cls

    def _plane_fused(self):
        """A 3-D transform larger than a pipeline chunk whose two leading passes are ROW x and COL y (what _slab_passes == 2 says) may
        run them as ONE persistent 2-D launch over its planes instead of slab by slab (round 6).  PYFFT_AMD_NO_PLANE_FUSED: never."""
        p = self._params
        # (transforms of up to half the cache -- one slab -- run their whole chain chunk-wise at the same rate: 0.244-0.265 either way,
        # profiles/r06_i_plane_fused_probe.log)
        return self._slab_passes == 2 and int(p.z) > 1 and p.size * p.complex_nbytes > self._context.machine.slab_bytes and \
            not D.no_plane_fused()

    def _persistent_rule(self):
        """The tuning rule of this plan's persistent launch (None: it has none) under the development switches in force."""
        narrow = N.lib.mifft_debug_get(N.DEBUG_NARROW_TILES) == 1
        return self._context.machine.tuning.match(self._shape_class(), narrow, not D.no_split_rowfirst())

    def _fused2d_eligible(self):
        rule = self._persistent_rule()
        return rule is not None and rule["kind"] == "2d"

    def _persistent_strategy(self, batch, forced):
        """(strategy, lag, ring, grid) of the rule's persistent launch for this batch, or None: the rule is on request only, the
        transform is below its admission size, the cache holds no ring, or the batch cannot fill the ring twice."""
        rule = self._persistent_rule()
        if rule is None or forced not in ("auto", "fused") or (rule.get("on_request") and forced != "fused"):
            return None
        p = self._params
        mach = self._context.machine
        rr = mach.tuning.ring_rule
        cls = self._shape_class()
        item_bytes = p.size * p.complex_nbytes
        name = rule["strategy"]
        planes = int(cls.get("planes", 1)) if rule["kind"] == "2d" else 1
        if planes > 1:
            # the work list's "transforms" are the (y, x) planes of the batch
            if name != "fused2" or rule.get("on_request"):
                return None
            item_bytes //= planes
            batch *= planes
        tiles0 = max(1, int(cls[rule["extent0"]]) // int(rule["cols0"]))
        per_cu = D.fused_grid_per_cu(int(rule["per_cu"]))
        geo = mach.fused_geometry(item_bytes, tiles0, per_cu, fill_cache=rule.get("ring") == "cache", min_slots=rule.get("min_slots"))
        if geo is None:
            return None
        lag, ring, grid = geo
        if name == "fused2":
            if int(rule["per_cu"]) == 1:
                lag, ring = D.fused3_lag_ring(lag, ring)
            lag, ring = D.fused_ring(lag, ring)
        # a batch that cannot fill the ring twice: halve the pipeline (2^18 x 160: 28 / 56 instead of 56 / 112) rather than fall back
        # to the chunks -- down to the 14 slots of the biggest transforms
        hn, hd = rr["halved_lag"][name]
        while forced == "auto" and batch < 2 * ring and ring >= rr["halve_while_ring_at_least"]:
            ring //= 2
            lag = max(1, hn * ring // hd)
        if name == "fusedp":
            lag, ring = D.fused_ring(lag, ring)
        elif forced == "fused" and batch < 2 * ring and batch >= 8:      # on request: a shorter pipeline for a small batch
            lag = batch // 4
            ring = 2 * lag
        big = item_bytes >= int(rule.get("min_item_bytes", 0)) or forced == "fused"
        return ("fused2z" if planes > 1 else name, lag, ring, grid) if (batch >= 2 * ring and big) else None

    def _development_strategy(self, batch, forced, tiny):
        """The measured-and-not-adopted forms, on request (PYFFT_AMD_* switches; `make DEV=1` builds of the library): the sequential
        list of a tiny batch, the per-XCD work lists, the XCD-resident single-crossing kernel.  docs/strategies.md."""
        p = self._params
        mach = self._context.machine
        dev = mach.tuning.development
        rule = self._persistent_rule()
        item_bytes = p.size * p.complex_nbytes
        if tiny:
            if D.small_fused(int(dev["small_fused_lag_div"])) and N.lib.mifft_has_feature(N.FEATURE_SEQUENTIAL_LIST) == 1 and \
                    not p.split and rule is not None and not rule.get("on_request") and not self._plane_fused():
                # statically dealt list: every work-group must be resident (1 or 2 per CU by the kernels' resources)
                return (rule["strategy"], 0, batch, int(rule["per_cu"]) * mach.compute_units)
            return None
        one_d = rule is not None and rule["kind"] == "1d" and p.precision == N.F32
        k = self._kernels
        xcd_ok = one_d and k[0].L <= 1024 and k[0].L >= k[1].L and mach.xcd_cooperative
        lists_ok = xcd_ok and N.lib.mifft_has_feature(N.FEATURE_FUSED2X) == 1
        if forced == "fusedx" and lists_ok and batch >= 64:
            lag, ring = D.fusedx()[:2]            # explicit lag / ring slots per XCD
            return ("fused2x", lag, ring, 2 * mach.compute_units)
        if forced == "xcd" and xcd_ok and k[0].L == 1024 and k[1].L == 1024 and not self._xcd2_disabled and \
                N.lib.mifft_has_feature(N.FEATURE_XCD2) == 1 and batch >= int(dev["xcd2_min_batch"]):
            return ("xcd2", D.xcd2_flags(N.XCD2_PREFETCH))
        if forced == "auto" and lists_ok and not D.no_fusedx():
            lag, ring = dev["fusedx_lag_ring"]
            # 2^17 on the 16-column tiles (MIFFT_NARROW_TILES=1): + 2 points over the pipelined chunks (profiles/r04_d_list_sweep.log)
            if k[0].L * k[1].L == (1 << 17) and N.lib.mifft_debug_get(N.DEBUG_NARROW_TILES) == 1 and \
                    batch >= 8 * 2 * ring and 8 * ring * item_bytes <= mach.ring_bytes:
                return ("fused2x", lag, ring, 2 * mach.compute_units)
            # split-complex planes with PYFFT_AMD_SPLIT_FUSEDX: sibling tiles share an XCD's L2 (profiles/r04_ac_split_siblings.log)
            if p.split and D.split_fusedx():
                ring = min(ring, (mach.ring_bytes * 4 // 7) // (8 * item_bytes))
                if ring >= 6 and batch >= 8 * 2 * ring:
                    return ("fused2x", ring // 2, ring, 2 * mach.compute_units)
        return None

    def _select_strategy(self, batch):
        """How the batch is cut and overlapped (docs/strategies.md).  Every measured constant comes from the tuning table
        (pyfft_amd/tuning_gfx950.json) and every size from the device (pyfft_amd/machine.py)."""
        forced = D.forced_strategy()
        p = self._params
        mach = self._context.machine
        item_bytes = p.size * p.complex_nbytes
        target = D.pipeline_chunk_bytes(mach.pipeline_chunk_bytes)
        # up to the cache size per side one launch per pass over the whole batch is the fastest form: the fork / join of the chunks and
        # the fill / drain of the persistent kernels only pay beyond it (profiles/r03_d_pipeline_threshold.log)
        tiny = forced == "auto" and batch * item_bytes <= mach.chain_max_bytes
        small = tiny or target < 1
        strat = self._development_strategy(batch, forced, tiny) if (tiny or not small) else None
        if strat is None and not small:
            strat = self._persistent_strategy(batch, forced)
        if strat is not None:
            return strat
        # any multi-pass plan gains from cache-sized chunks (the second pass re-reads what the first just wrote)
        chunk = max(1, target // item_bytes) if target >= 1 else 1
        pipe = mach.tuning.pipelined
        if not small and len(self._kernels) >= 2 and forced in ("auto", "pipelined") and batch >= pipe["min_chunks"] * chunk:
            nslab = 0
            if self._slab_passes and not D.no_slabs():
                plane_bytes = int(p.x) * int(p.y) * p.complex_nbytes
                slab_target = D.slab_bytes(mach.slab_bytes // (pipe["split_slab_divisor"] if p.split else 1))   # split planes measured best at half
                nslab = int(p.z) // min(int(p.z), max(1, slab_target // plane_bytes))
            return ("pipelined", chunk, D.pipeline_streams(pipe["streams"]), nslab)
        return ("chain",)

    PERSISTENT = ("fused2", "fused2x", "fusedp", "fused2z")

    def _prepare(self, batch):
        """Choose the strategy when the batch changes (plan.py:179-192); the plan-owned scratch of that strategy is allocated by the
        first execute that needs it (_ensure_scratch): an out-of-place execute that takes the one-launch route (_runs_oop_nd) needs none."""
        if self._last_batch_size == batch:
            return
        self._retain_captured_scratch()
        self._last_batch_size = batch
        self._last_call_key = None
        self._strategy = self._select_strategy(batch)
        self._tempmemobj = None
        self._scratch_ready = False

    def _retain_captured_scratch(self):
        """A captured launch holds raw device addresses of this plan's scratch (ring / temp buffer, counter sets, pinned error word,
        side streams): once an execute() has been recorded into a graph, scratch that is about to be replaced or released moves to
        a keep-alive list instead of being freed, so that later replays of that graph still find it (hip.Graph docstring)."""
        if self._captured:
            held = (self._tempmemobj, self._counters, self._errword, self._xcd2_scratch, self._side_streams, self._side_events)
            if any(h is not None for h in held) and not any(k is held or k == held for k in self._capture_keepalive):
                self._capture_keepalive.append(held)

    def _scratch_needed(self):
        return self._strategy[0] == "xcd2" or self._temp_buffer_needed or self._strategy[0] in self.PERSISTENT or \
            (self._strategy[0] == "pipelined" and self._side_streams is None)

    def _ensure_scratch(self):
        """(Re)allocate the plan-owned scratch of the current batch's strategy (plan.py:179-192)."""
        if self._scratch_ready:
            return
        ctx = self._context
        p = self._params
        batch = self._last_batch_size
        if self._scratch_needed() and ctx.capturing():
            # scratch, counters and side streams are allocated here, which a capturing stream cannot record
            raise RuntimeError("pyfft_amd: execute() on a capturing stream needs one eager execute() of the same batch (and the same "
                               "in-place / out-of-place form) first")
        self._scratch_ready = True
        if self._strategy[0] == "pipelined" and self._side_streams is None:
            from .hip import Stream, Event
            self._side_streams = [Stream() for _ in range(self._strategy[2])]
            self._side_events = [Event() for _ in range(self._strategy[2] + 1)]
        if self._strategy[0] == "xcd2":
            if self._xcd2_scratch is None:
                self._xcd2_scratch = ctx.allocate_raw(N.XCD2_SCRATCH_BYTES)
            self._counters = ctx.allocate_raw(N.XCD2_CONTROL_BYTES + N.XCD2_TRACE_BYTES)
            return
        if not self._temp_buffer_needed and self._strategy[0] not in self.PERSISTENT:
            return
        if self._strategy[0] in self.PERSISTENT:
            # ring slots: `ring` of them; the per-XCD lists hold `ring` slots for each of the 8 XCDs
            items = self._strategy[2] * (8 if self._strategy[0] == "fused2x" else 1)
            # two counter sets: every launch runs on one and zeroes the other (mifft_fused_sync), so no memset precedes a launch;
            # a third one for launches captured into a graph (_fused_sync)
            planes = int(p.z) if self._strategy[0] == "fused2z" else 1
            self._counter_bytes = N.fused2_counter_bytes(batch * planes)
            self._counters = ctx.allocate_raw(3 * self._counter_bytes)
            self._counters_clean = False
            if self._errword is None:
                from .hip import ErrorWord
                self._errword = ErrorWord()
        elif self._strategy[0] == "pipelined":
            items = self._strategy[1] * self._strategy[2]  # chunk * streams
        else:
            items = batch
        # one interleaved buffer for both layouts (the reference allocates two scalar planes for split plans,
        # plan.py:189-190; same total size)
        item = p.size // int(p.z) if self._strategy[0] == "fused2z" else p.size      # fused2z: the ring holds (y, x) planes
        self._tempmemobj = ctx.allocate(item * items * p.complex_nbytes)

    def _fused_sync(self, stream, capturing=False):
        """mifft_fused_sync of the coming persistent launch: the counter set it runs on (zero: the previous launch cleared it, or
        the memset below), the set it clears for the next launch, the pinned error word.  A launch that is being CAPTURED into a
        graph replays on the same set every time, so it takes the third set in the single-set form -- the library zeroes it with a
        memset node in front of the kernel -- and leaves the alternation of the eager launches alone."""
        base = self._context.pointer_of(self._counters)
        nb = self._counter_bytes
        if capturing:
            return N.MifftFusedSync(base + 2 * nb, None, self._errword.ptr)
        if D.fused_memset():            # development A/B: one counter set, zeroed by a memset in front of every launch
            return N.MifftFusedSync(base, None, self._errword.ptr)
        if not self._counters_clean:
            N.check(N.lib.mifft_memset(base, 0, 2 * nb, stream), "mifft_memset")
            self._counters_clean = True
            self._counter_set = 0
        cur = self._counter_set
        self._counter_set = 1 - cur
        return N.MifftFusedSync(base + cur * nb, base + (1 - cur) * nb, self._errword.ptr)

    def _runs_oop_nd(self, batch):
        """An OUT-OF-PLACE execute of this batch takes the one-launch kernel with several work-groups per transform (csrc/fft_nd2z.hpp)
        instead of the plan's chain / strategy: always for the shapes whose two halves run on two-per-CU tiles, beyond half the
        last-level cache per side for the four-quarter shapes."""
        return self._oop_nd is not None and not D.no_oop_nd() and D.forced_strategy() == "auto" and \
            (self._oop_any_size or batch * self._params.size * self._params.complex_nbytes > self._context.machine.write_through_max_bytes)

    def _enqueue(self, batch, is_inplace, inverse, bufs0, bufs1, capturing=False):
        ctx = self._context
        stream = ctx.stream_handle()
        if not is_inplace and self._runs_oop_nd(batch):
            # one launch, several work-groups per transform (csrc/fft_nd2z.hpp); needs no scratch.  Beyond half the last-level cache per
            # side: (256, 256) at 256 MiB 0.339 -> 0.444, at 2 GiB 0.464 (persistent) -> 0.501; at 32 MiB the two launches win (0.404 / 0.356)
            descs = self._descriptors(batch, False, bool(inverse), alt=2)
            N.check(N.lib.mifft_launch_chain(descs, 1, bufs0, bufs1, stream), "mifft_launch_chain")
            return
        descs = self._descriptors(batch, is_inplace, bool(inverse))
        strat = self._strategy
        if strat[0] == "xcd2":
            if capturing:
                raise RuntimeError("pyfft_amd: the development strategy xcd2 cannot be captured into a graph")
            d0, d1 = descs[0], descs[1]
            in1 = bufs1[d0.src] if bufs1 is not None else None
            out1 = bufs1[d1.dst] if bufs1 is not None else None
            N.check(N.lib.mifft_launch_xcd2(ctypes.byref(d0), ctypes.byref(d1), bufs0[d0.src], in1, bufs0[d1.dst], out1,
                                            ctx.pointer_of(self._xcd2_scratch), ctx.pointer_of(self._counters),
                                            strat[1], stream), "mifft_launch_xcd2")
            self._post_error_word(stream)
        elif strat[0] in self.PERSISTENT:
            sync = self._fused_sync(stream, capturing)
            try:
                if strat[0] == "fused2x":
                    _, lag, ring, grid = strat
                    d0, d1 = descs[0], descs[1]
                    in1 = bufs1[d0.src] if bufs1 is not None else None
                    out1 = bufs1[d1.dst] if bufs1 is not None else None
                    N.check(N.lib.mifft_launch_fused2x(ctypes.byref(d0), ctypes.byref(d1), bufs0[d0.src], in1, bufs0[d1.dst], out1, bufs0[2],
                                                       ring, lag, ctypes.byref(sync), grid, stream), "mifft_launch_fused2x")
                elif strat[0] == "fusedp":
                    _, lag, ring, grid = strat
                    if self._pair_alt is not None:       # the four-pass list of this launch alone (the chain is plane pass + z pass)
                        descs = self._descriptors(batch, is_inplace, bool(inverse), alt=True)
                    in1 = bufs1[descs[0].src] if bufs1 is not None else None
                    out1 = bufs1[descs[3].dst] if bufs1 is not None else None
                    N.check(N.lib.mifft_launch_fused_pair(descs, bufs0[descs[0].src], in1, bufs0[descs[3].dst], out1, bufs0[2], ring, lag,
                                                          ctypes.byref(sync), grid, stream), "mifft_launch_fused_pair")
                else:
                    _, lag, ring, grid = strat
                    d0, d1 = descs[0], descs[1]
                    # the two-pass schedule is in -> temp -> out for both in-place and out-of-place calls
                    in1 = bufs1[d0.src] if bufs1 is not None else None
                    out1 = bufs1[d1.dst] if bufs1 is not None else None
                    N.check(N.lib.mifft_launch_fused2(ctypes.byref(d0), ctypes.byref(d1), bufs0[d0.src], in1, bufs0[d1.dst], out1,
                                                      bufs0[2], None, ring, lag, ctypes.byref(sync), grid, stream),
                            "mifft_launch_fused2")
                    if strat[0] == "fused2z":
                        # the (y, x) planes are done where the chain's second pass leaves them; the z passes as plain launches
                        rest = ctypes.cast(ctypes.byref(descs, 2 * ctypes.sizeof(N.MifftPass)), ctypes.POINTER(N.MifftPass))
                        N.check(N.lib.mifft_launch_chain(rest, len(self._kernels) - 2, bufs0, bufs1, stream), "mifft_launch_chain")
            except Exception:
                self._counters_clean = False     # a launch that did not start cleared nothing
                raise
        elif strat[0] == "pipelined":
            _, chunk, nside, nslab = strat
            if capturing:
                # recorded into a graph: the chunks one after the other on the capturing stream itself (the library skips the fork / join
                # when the only "side" stream IS the caller's) -- the same launches on the same chunks, hence the same bits, as a LINEAR
                # graph.  A forked capture replayed correctly most of the time and took the process down in hipGraphLaunch once in
                # five runs (ROCm 7.2: profiles/r05_capture_pipelined_crash.log), and a replayed graph has no launch gaps to hide anyway.
                nside = 1
                side = (ctypes.c_void_p * 1)(stream)
            else:
                side = (ctypes.c_void_p * nside)(*[s.handle for s in self._side_streams])
            evs = (ctypes.c_void_p * (nside + 1))(*[e.handle for e in self._side_events[:nside + 1]])
            npass = len(self._kernels)
            first = 0
            if nslab > 1:
                # leading x / y passes slab by slab: "batch" = batch * nslab slabs of size / nslab points, one per chunk
                first = self._slab_passes
                N.check(N.lib.mifft_launch_chain_pipelined(descs, first, bufs0, bufs1, batch * nslab, 1,
                                                           self._params.size // nslab, stream, side, nside, evs),
                        "mifft_launch_chain_pipelined")
            rest = ctypes.cast(ctypes.byref(descs, first * ctypes.sizeof(N.MifftPass)), ctypes.POINTER(N.MifftPass))
            N.check(N.lib.mifft_launch_chain_pipelined(rest, npass - first, bufs0, bufs1, batch, chunk,
                                                       self._params.size, stream, side, nside, evs),
                    "mifft_launch_chain_pipelined")
        else:
            N.check(N.lib.mifft_launch_chain(descs, len(self._kernels), bufs0, bufs1, stream), "mifft_launch_chain")

    # ---- error reporting of the persistent kernels --------------------------------------------------------------------
    # fused2 / fused2x / fusedp write a PINNED HOST word (hip.ErrorWord) when a bounded dependency wait times out; the host looks
    # at it on entry of every execute() (check(), never blocks) and after synchronising (finish()): an asynchronous caller learns
    # of invalid results at its next call at the latest.  The development strategy xcd2 keeps rounds 2-3's mechanism: word [1] of
    # its control block copied into pinned memory behind every launch (hip.ErrorMailbox).
    def _post_error_word(self, stream):
        if self._mailbox is None:
            from .hip import ErrorMailbox
            self._mailbox = ErrorMailbox()
        self._mailbox.post(self._context.pointer_of(self._counters) + 4, stream, self._strategy[0])

    def _take_error_word(self):
        if self._errword is not None:
            word = self._errword.take()
            if word:
                self._counters_clean = False
                raise RuntimeError("pyfft_amd: %s kernel dependency time-out (results invalid)" % (self._strategy[0],))

    @on_plan_device
    def check(self):
        """Raise if a completed asynchronous execute() reported invalid results (non-blocking)."""
        self._take_error_word()
        if self._mailbox is not None:
            self._handle_errors(self._mailbox.collect(False))

    @on_plan_device
    def finish(self):
        """Wait for the plan's stream, then raise if any execute() since the last check reported invalid results."""
        self._context.wait()
        self._take_error_word()
        if self._mailbox is not None:
            self._handle_errors(self._mailbox.collect(True))

    def _handle_errors(self, errors):
        for strategy, word in errors:
            if strategy == "xcd2" and (word & 2):
                # the launch did not find 64 resident work-groups per XCD (the device is shared): nothing was written;
                # this plan stops using the strategy
                self._xcd2_disabled = True
                self._last_batch_size = 0
                raise RuntimeError("pyfft_amd: XCD-cooperative launch found no full XCD residency; results of that execute() "
                                   "are invalid -- the plan has switched strategy, run it again")
            raise RuntimeError("pyfft_amd: %s kernel dependency time-out (results invalid)" % strategy)

    def _buffers(self, is_inplace, args):
        ptr = self._context.pointer_of
        if self._params.split:
            in_re, in_im, out_re, out_im = (ptr(a) for a in args)
            # explicit aliasing is an in-place call (Appendix A item 5 of SURVEY.md)
            if not is_inplace and (in_re == out_re or in_im == out_im):
                if in_re == out_re and in_im == out_im:
                    is_inplace = True
                else:
                    raise ValueError("partially aliased split buffers")
            bufs0 = N.make_buf3(in_re, out_re, ptr(self._tempmemobj) if self._tempmemobj is not None else None)
            bufs1 = N.make_buf3(in_im, out_im, None)
        else:
            d_in, d_out = (ptr(a) for a in args)
            if not is_inplace and d_in == d_out:
                is_inplace = True
            bufs0 = N.make_buf3(d_in, d_out, ptr(self._tempmemobj) if self._tempmemobj is not None else None)
            bufs1 = None
        return is_inplace, bufs0, bufs1

    @on_plan_device
    def _execute(self, wait_for_finish, is_inplace, inverse, batch, *args):
        """Execute plan for given data type (plan.py:173-259)."""
        ctx = self._context
        batch = int(batch)
        if batch < 1:
            raise ValueError("batch must be positive")
        if self._mailbox is not None or self._errword is not None:
            self.check()
        ctx.createQueue(args)
        # global wait setting has lower priority than the local one (plan.py:250-253)
        wait = self._wait_for_finish
        if wait_for_finish is not None:
            wait = wait_for_finish
        capturing = ctx.capturing()
        if capturing and wait:
            # finish() synchronises the stream, which a capturing stream refuses and which invalidates the capture
            raise RuntimeError("pyfft_amd: execute() on a capturing stream cannot wait for the result: build the plan with stream= "
                               "(or wait_for_finish=False), or pass wait_for_finish=False to this call")
        if self._last_batch_size != batch:
            self._prepare(batch)
        # small transforms are launch-bound (a 32 MiB execute is ~12 us of device time): the pointer triples of the last
        # call are kept, so that repeated executes on the same buffers skip rebuilding them
        ptr = ctx.pointer_of
        key = (is_inplace,) + tuple(ptr(a) for a in args)
        if key != self._last_call_key:
            self._last_call = self._buffers(is_inplace, args)
            self._last_call_key = key
        is_inplace, bufs0, bufs1 = self._last_call
        if not self._scratch_ready and (is_inplace or not self._runs_oop_nd(batch)):
            self._ensure_scratch()
            if self._tempmemobj is not None:
                self._last_call = self._buffers(is_inplace, args)      # (the triple carries the temp buffer's address)
                is_inplace, bufs0, bufs1 = self._last_call

        ctx.order_scratch(capturing)
        if capturing:
            self._note_capture()
        self._enqueue(batch, is_inplace, inverse, bufs0, bufs1, capturing)

        if wait:
            self.finish()
        else:
            ctx.flush()
            return ctx.getQueue()

    def _note_capture(self):
        """This execute() is being recorded into a graph: the graph bakes in the addresses of the plan's scratch and tables.  The plan
        keeps that scratch alive for as long as it lives itself (_retain_captured_scratch), and an open hip.Graph takes a reference to
        the plan, so that `del plan` with the graph still around frees nothing the graph replays on."""
        self._captured = True
        from .hip import Graph
        Graph.retain(self)

    def _executeInterleaved(self, data_in, data_out=None, inverse=False, batch=1, wait_for_finish=None):
        """Execute plan for interleaved complex array (plan.py:261-271)."""
        if data_out is None:
            data_out = data_in
            is_inplace = True
        else:
            is_inplace = False
        return self._execute(wait_for_finish, is_inplace, inverse, batch, data_in, data_out)

    def _executeSplit(self, data_in_re, data_in_im, data_out_re=None, data_out_im=None,
                      inverse=False, batch=1, wait_for_finish=None):
        """Execute plan for split complex array (plan.py:273-284)."""
        if data_out_re is None and data_out_im is None:
            data_out_re = data_in_re
            data_out_im = data_in_im
            is_inplace = True
        else:
            if data_out_re is None or data_out_im is None:
                raise ValueError("both output planes must be given")
            is_inplace = False
        return self._execute(wait_for_finish, is_inplace, inverse, batch,
                             data_in_re, data_in_im, data_out_re, data_out_im)

    def close(self):
        """Wait for outstanding work and release the plan's device resources now (temp buffer, counters, scratch, side
        streams and events) instead of at garbage collection.  The plan stays usable: everything is re-created on demand.
        Scratch that a captured graph replays on (an execute() of this plan was recorded by hip.Graph / torch.cuda.graph()) is NOT
        released here: it lives until the plan is collected, or until release_captured() says that no such graph will run again."""
        try:
            self.finish()
        finally:
            self._retain_captured_scratch()      # (a plan that was captured into a graph keeps what the graph replays on)
            self._scratch_ready = False
            self._tempmemobj = None
            self._counters = None
            self._xcd2_scratch = None
            self._side_streams = None
            self._side_events = None
            self._mailbox = None
            self._errword = None
            self._last_batch_size = 0
            self._last_call_key = None

    def release_captured(self):
        """The caller guarantees that no graph recorded from this plan will be launched again: waits for the plan's stream and frees
        the scratch kept alive for such graphs (replaced rings, counter sets, error words)."""
        self.finish()
        self._capture_keepalive = []
        self._captured = False

    # ------------------------------------------------------------------------------------
    # introspection helpers used by bench.py / tests (not part of the reference API)
    def pass_list(self, inplace=True):
        """The plan's pass chain; inplace=False: what an OUT-OF-PLACE execute launches where that differs (the one-pass list of the
        shapes with a several-work-groups-per-transform kernel, csrc/fft_nd2z.hpp -- whether a given batch takes it: strategy())."""
        if not inplace and self._oop_nd is not None and not D.no_oop_nd() and D.forced_strategy() == "auto":
            return list(self._oop_nd)
        return list(self._kernels)

    @on_plan_device
    def strategy(self, batch, inplace=True):
        """(name, ...) of the execution strategy of this batch.  inplace=False: what an out-of-place execute runs -- ("nd_oop",)
        where it bypasses the chain for the one-launch kernel (_runs_oop_nd; needs no scratch), else the same as in place."""
        self._prepare(int(batch))
        if not inplace and self._runs_oop_nd(int(batch)):
            return ("nd_oop",)
        return self._strategy

    @on_plan_device
    def timed_execute(self, repeats, is_inplace, inverse, batch, bufs_in, bufs_out):
        """Device time (ms) of `repeats` back-to-back executions, measured with HIP events recorded on the
        plan's stream (hipEvents see the stream the kernels are launched on)."""
        from .hip import Event
        ctx = self._context
        batch = int(batch)
        self._prepare(batch)
        args = (bufs_in[0], bufs_in[1], bufs_out[0], bufs_out[1]) if self._params.split else (bufs_in[0], bufs_out[0])
        if is_inplace or not self._runs_oop_nd(batch):
            self._ensure_scratch()
        is_inplace, bufs0, bufs1 = self._buffers(is_inplace, args)
        ctx.createQueue()
        e0, e1 = Event(), Event()
        e0.record(ctx.getQueue())
        for _ in range(int(repeats)):
            self._enqueue(batch, is_inplace, inverse, bufs0, bufs1)
        e1.record(ctx.getQueue())
        e1.synchronize()
        self.finish()
        return e1.time_since(e0)

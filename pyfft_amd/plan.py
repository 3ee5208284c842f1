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
        self._counters = None
        self._xcd2_scratch = None
        self._xcd2_disabled = False
        self._mailbox = None
        self._last_call_key = None
        self._last_call = None
        self._side_streams = None
        self._side_events = None

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
        if int(p.z) > 1 and not self._temp_buffer_needed and not self._paired and p.size * p.complex_nbytes > self.PIPELINE_TARGET_BYTES:
            k = 0
            while k < len(self._kernels) and self._kernels[k].kind != N.PASS_ND and \
                    self._kernels[k].axis in (P.X_DIRECTION, P.Y_DIRECTION):
                k += 1
            if 2 <= k < len(self._kernels):
                self._slab_passes = k

        self._tables = {}      # key -> device allocation
        self._table_ptrs = []  # per pass: (tw_L, tw_lo, tw_hi, shift)
        for k in self._kernels:
            if k.kind == N.PASS_ND:
                # one w(len)^k table per axis (x, y, z) in the tw_L / tw_lo / tw_hi slots
                tabs = [self._device_table(("L", n), lambda L=n: _twiddle_table(L, L, 1, p.complex_dtype)) if n > 1 else None
                        for n in (k.L, k.M, k.S)]
                self._table_ptrs.append((tabs[0], tabs[1], tabs[2], 0))
                continue
            twL = self._device_table(("L", k.L), lambda L=k.L: _twiddle_table(L, L, 1, p.complex_dtype))
            if k.M > 1:
                n = k.curr_n
                shift = (P.log2(n) + 1) // 2
                lo = self._device_table(("lo", n, shift),
                                        lambda n=n, s=shift: _twiddle_table(n, 1 << s, 1, p.complex_dtype))
                hi = self._device_table(("hi", n, shift),
                                        lambda n=n, s=shift: _twiddle_table(n, n >> s, 1 << s, p.complex_dtype))
                self._table_ptrs.append((twL, lo, hi, shift))
            else:
                self._table_ptrs.append((twL, None, None, 0))

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

    def _descriptors(self, batch, is_inplace, inverse):
        """Pass descriptor array for (batch, schedule, direction); cached like the reference's
        kernel.prepare(batch) (kernel.py:85-93)."""
        key = (batch, is_inplace, inverse)
        d = self._desc_cache.get(key)
        if d is not None:
            return d
        p = self._params
        _, sched = P.buffer_schedule(self._kernels, is_inplace, self._via_temp)
        arr = (N.MifftPass * max(1, len(self._kernels)))()
        last = len(self._kernels) - 1
        for i, (k, (src, dst), (twL, lo, hi, shift)) in enumerate(zip(self._kernels, sched, self._table_ptrs)):
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
            if last >= 1 and (p.size * p.complex_nbytes <= (64 << 20) or self._slab_passes or self._paired) and not D.no_stream_hints():
                if i == 0:
                    d.flags |= N.FLAG_STREAM_SRC
                if i == last:
                    d.flags |= N.FLAG_STREAM_DST
            # one-launch N-D plans on buffers beyond the Infinity Cache: non-temporal stores of the result (nobody finds it in a
            # cache anyway): (16, 16, 16) 70.4 -> 72.6 %, (128, 128) 64.8 -> 68.1 %, fp64 (128, 128) 58.6 -> 59.9 % at 1 GiB
            # (profiles/r03_c_store_policy.log; the long ROW kernels measured better with plain stores and keep them)
            if last == 0 and k.kind == N.PASS_ND and batch * p.size * p.complex_nbytes > self.CHAIN_MAX_BYTES and not D.no_stream_hints():
                d.flags |= N.FLAG_STREAM_DST
            # small launches: write-through stores, so that the output does not wait dirty in the L2s for the end-of-kernel
            # write-back (32 MiB launches: (16, 16, 16) 14.3 -> 9.1 us, (1024,) 13.5 -> 10.5 us; neutral from 256 MiB on)
            if batch * p.size * p.complex_nbytes <= self.WRITE_THROUGH_MAX_BYTES and not D.no_stream_hints():
                d.flags = (d.flags & ~N.FLAG_STREAM_DST) | N.FLAG_WRITE_THROUGH
        if len(self._desc_cache) > 64:
            self._desc_cache.clear()
        self._desc_cache[key] = arr
        return arr

    # ------------------------------------------------------------------------------------
    # ------------------------------------------------------------------------------------
    # execution strategies (all enqueue the same passes; they differ in how the batch is cut and overlapped)
    #   chain      one launch per pass over the whole batch (the reference's loop, plan.py:217-248)
    #   pipelined  batch cut into Infinity-Cache-sized chunks, chunk i on side stream i % n with its own temp
    #              slot (mifft_launch_chain_pipelined)
    #   fused2     both passes of a long 1-D fp32 transform in one persistent launch (mifft_launch_fused2)
    #   xcd2       1024 x 1024 fp32: one persistent launch, each transform stays on one XCD between its two HBM
    #              crossings (mifft_launch_xcd2); needs no temp buffer, in place or out of place
    PIPELINE_TARGET_BYTES = 64 << 20
    SLAB_TARGET_BYTES = 128 << 20      # slabs of the leading passes of a big 3-D transform (C4: 24.7 % at 64 MiB, 25.5 % at 128)
    PIPELINE_STREAMS = 2
    CHAIN_MAX_BYTES = 256 << 20        # per side (batch x transform): below this the plain launch chain wins
    WRITE_THROUGH_MAX_BYTES = 128 << 20   # per side: below this every launch stores write-through (_descriptors)
    XCD2_MIN_BATCH = 64                # 8 transforms per XCD: below that the pipelined chunks win
    SMALL_FUSED_LAG_DIV = 0            # small-batch fused form: off (see _select_strategy)

    def _fused2d_eligible(self):
        """2-D 1024 x 1024 (BASELINE config 3, and the published double-precision shape): ROW + strided COL, run by the fused
        kernel as two transposing passes."""
        p = self._params
        k = self._kernels
        # (split planes: 29 % against 35 % for the pipelined chunks -- only on request)
        side = int(p.x)
        if side not in ((1024,) if p.precision == N.F64 else (512, 1024, 2048)):
            return False
        return (int(p.y) == side and int(p.z) == 1 and len(k) == 2
                and (not p.split or D.forced_strategy() == "fused")
                and k[0].kind == N.PASS_ROW and k[0].L == side and k[1].kind == N.PASS_COL and k[1].L == side
                and k[1].M == 1 and k[1].S == side)

    def _fused2_eligible(self):
        p = self._params
        k = self._kernels
        if self._fused2d_eligible():
            return True
        if not (len(k) == 2 and int(p.y) == 1 and int(p.z) == 1 and k[0].kind == N.PASS_COL and k[1].kind == N.PASS_COL
                and k[0].S == 1 and k[0].M == k[1].L and k[1].M == 1):
            return False
        if p.precision == N.F64:
            return k[0].L == 1024 and k[1].L == 1024
        return (k[0].L in (256, 512, 1024) and k[1].L in (256, 512, 1024)) or (k[0].L == 2048 and k[1].L in (1024, 2048))

    def _xcd2_eligible(self):
        k = self._kernels
        return (self._fused2_eligible() and not self._fused2d_eligible() and self._params.precision == N.F32
                and k[0].L == 1024 and k[1].L == 1024 and self._context.compute_units == 256
                and not self._xcd2_disabled)

    def _select_strategy(self, batch):
        forced = D.forced_strategy()
        p = self._params
        item_bytes = p.size * p.complex_nbytes
        target = D.pipeline_chunk_bytes(self.PIPELINE_TARGET_BYTES)
        nstreams = D.pipeline_streams(self.PIPELINE_STREAMS)
        chunk = max(1, target // item_bytes)
        strat = ("chain",)
        # up to 256 MiB per side one launch per pass over the whole batch is the fastest form: the side-stream fork / join of the
        # pipelined chunks and the fill / drain of the persistent kernels only pay beyond it (profiles/r03_d_pipeline_threshold.log:
        # (1024, 1024) x 32 chain 0.348 / pipelined 0.317, 2^18 x 128 chain 0.374 / fused 0.329, 2^16 x 512 0.387 / 0.343,
        # 128^3 x 16 0.346 / 0.299; at twice the size the order flips)
        if forced == "auto" and batch * item_bytes <= self.CHAIN_MAX_BYTES:
            return strat
        if forced == "fusedx" and self._fused2_eligible() and not self._fused2d_eligible() and not p.split and p.precision == N.F32 \
                and self._kernels[0].L <= 1024 and batch >= 64:
            lag, ring, wt = D.fusedx()            # development: one work list per XCD, ring slots per XCD
            return ("fused2x", lag, ring, 2 * self._context.compute_units, wt)
        if forced == "xcd" and self._xcd2_eligible() and batch >= self.XCD2_MIN_BATCH:   # not the default yet: DESIGN.md section 4
            return ("xcd2", D.xcd2_flags(N.XCD2_PREFETCH))
        if (self._temp_buffer_needed or self._fused2d_eligible()) and forced in ("auto", "fused") and self._fused2_eligible():
            huge = self._kernels[1].L == 2048 or p.precision == N.F64   # 512-thread tiles: one work-group per CU
            grid = D.fused_grid_per_cu(1 if huge else 2) * self._context.compute_units   # (four per CU for L <= 512: no gain)
            gsize = 2 * max(max(self._kernels[0].M, self._kernels[0].L if self._fused2d_eligible() else 1) // 16, self._kernels[1].S // 16)
            # producers run `lag` transforms ahead of the consumers; ring = 2 * lag slots (1024 x 1024: lag 14, 224 MiB --
            # the largest ring that still fits the 256 MiB Infinity Cache measured best: fused_probe.py wide)
            # measured on MI355X (end of round 2, counters on their own lines): the persistent kernel beats the stream-pipelined
            # chunks from N = 2^18 up (2^18: 42.0 vs 39.5 %, 2^19: 37.2 vs 36.2 %; 2^17: 39.1 vs 39.5, 2^16: 33 vs 40)
            big = self._kernels[0].L * self._kernels[1].L >= (1 << 18)
            lag = max(2, -(-D.fused_lag_factor(14) * grid // (4 * gsize)))
            ring = 2 * lag
            if huge:
                # fp32 2048 x 2048: 32 MiB per transform, fp64 1024 x 1024: 16 MiB; the ring that fits the Infinity Cache is
                # 224 MiB, and the consumers follow the producers by a bit more than half of it (measured: lag 4 of 7 / 8 of 14)
                slots = (224 << 20) // item_bytes
                lag, ring = D.fused3_lag_ring(4 * slots // 7, slots)
                big = True
            lag, ring = D.fused_ring(lag, ring)
            if forced == "fused" and batch < 2 * ring and batch >= 8:   # on request: a shorter pipeline for a small batch
                lag = batch // 4
                ring = 2 * lag
            if batch >= 2 * ring and (big or forced == "fused"):
                return ("fused2", lag, ring, grid)
            # small batches (the reference's own 32 MiB protocol: (1024, 1024) x 4) through the same persistent launch with one ring
            # slot per transform and the consumers `lag` transforms behind: MEASURED SLOWER than one launch per pass at every
            # batch below the normal form's threshold ((1024, 1024) x 4: 64.9 against 28.4 us, x 16: 133 against 85 us; the
            # launch's counter reset + error-word copy cost ~30 us and a short pipeline never fills: profiles/
            # r03_small_batch_fused.log) -- development switch only (PYFFT_AMD_SMALL_FUSED = lag divisor), off by default
            small = D.small_fused(self.SMALL_FUSED_LAG_DIV)
            if small and 2 <= batch < 2 * ring and (big or forced == "fused") and forced != "pipelined" and \
                    (batch < 4 * chunk or forced == "fused"):
                return ("fused2", max(1, batch // small), batch, grid)
        # any multi-pass plan gains from cache-sized chunks (the second pass re-reads what the first just wrote),
        # whether or not it needs a temp buffer
        if len(self._kernels) >= 2 and forced in ("auto", "pipelined") and batch >= 4 * chunk:
            nslab = 0
            if self._slab_passes and not D.no_slabs():
                plane_bytes = int(p.x) * int(p.y) * p.complex_nbytes
                slab_target = D.slab_bytes(self.SLAB_TARGET_BYTES // 2 if p.split else self.SLAB_TARGET_BYTES)   # split planes measured best at 64 MiB
                planes = min(int(p.z), max(1, slab_target // plane_bytes))
                nslab = int(p.z) // planes
            return ("pipelined", chunk, nstreams, nslab)
        return strat

    def _prepare(self, batch):
        """(Re)allocate the plan-owned scratch when the batch changes (plan.py:179-192)."""
        ctx = self._context
        p = self._params
        if self._last_batch_size == batch:
            return
        self._last_batch_size = batch
        self._last_call_key = None
        self._strategy = self._select_strategy(batch)
        self._tempmemobj = None
        if self._strategy[0] == "pipelined" and self._side_streams is None:
            from .hip import Stream, Event
            self._side_streams = [Stream() for _ in range(self._strategy[2])]
            self._side_events = [Event() for _ in range(self._strategy[2] + 1)]
        if self._strategy[0] == "xcd2":
            if self._xcd2_scratch is None:
                self._xcd2_scratch = ctx.allocate_raw(N.XCD2_SCRATCH_BYTES)
            self._counters = ctx.allocate_raw(N.XCD2_CONTROL_BYTES + N.XCD2_TRACE_BYTES)
            return
        if not self._temp_buffer_needed and self._strategy[0] not in ("fused2", "fused2x"):
            return
        if self._strategy[0] == "fused2x":
            items = (1 if self._strategy[4] == 2 else 8) * self._strategy[2]     # ring slots per XCD (wt == 2: one global ring)
            self._counters = ctx.allocate_raw(N.fused2_counter_bytes(batch))
        elif self._strategy[0] == "fused2":
            items = self._strategy[2]                     # ring slots
            self._counters = ctx.allocate_raw(N.fused2_counter_bytes(batch))
        elif self._strategy[0] == "pipelined":
            items = self._strategy[1] * self._strategy[2]  # chunk * streams
        else:
            items = batch
        # one interleaved buffer for both layouts (the reference allocates two scalar planes for split plans,
        # plan.py:189-190; same total size)
        self._tempmemobj = ctx.allocate(p.size * items * p.complex_nbytes)

    def _enqueue(self, batch, is_inplace, inverse, bufs0, bufs1):
        ctx = self._context
        descs = self._descriptors(batch, is_inplace, bool(inverse))
        stream = ctx.stream_handle()
        strat = self._strategy
        if strat[0] == "xcd2":
            d0, d1 = descs[0], descs[1]
            in1 = bufs1[d0.src] if bufs1 is not None else None
            out1 = bufs1[d1.dst] if bufs1 is not None else None
            N.check(N.lib.mifft_launch_xcd2(ctypes.byref(d0), ctypes.byref(d1), bufs0[d0.src], in1, bufs0[d1.dst], out1,
                                            ctx.pointer_of(self._xcd2_scratch), ctx.pointer_of(self._counters),
                                            strat[1], stream), "mifft_launch_xcd2")
            self._post_error_word(stream)
        elif strat[0] == "fused2x":
            _, lag, ring, grid, wt = strat
            d0, d1 = descs[0], descs[1]
            N.check(N.lib.mifft_launch_fused2x(ctypes.byref(d0), ctypes.byref(d1), bufs0[d0.src], bufs0[d1.dst], bufs0[2], ring, lag,
                                               ctx.pointer_of(self._counters), grid, wt, stream), "mifft_launch_fused2x")
            self._post_error_word(stream)
        elif strat[0] == "fused2":
            _, lag, ring, grid = strat
            d0, d1 = descs[0], descs[1]
            # the two-pass schedule is in -> temp -> out for both in-place and out-of-place calls
            in1 = bufs1[d0.src] if bufs1 is not None else None
            out1 = bufs1[d1.dst] if bufs1 is not None else None
            N.check(N.lib.mifft_launch_fused2(ctypes.byref(d0), ctypes.byref(d1), bufs0[d0.src], in1, bufs0[d1.dst], out1,
                                              bufs0[2], None, ring, lag, ctx.pointer_of(self._counters), grid, stream),
                    "mifft_launch_fused2")
            self._post_error_word(stream)
        elif strat[0] == "pipelined":
            _, chunk, nside, nslab = strat
            side = (ctypes.c_void_p * nside)(*[s.handle for s in self._side_streams])
            evs = (ctypes.c_void_p * (nside + 1))(*[e.handle for e in self._side_events])
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

    # ---- error word of the persistent kernels (fused2 / xcd2): word [1] of their control block --------------------
    # Every launch is followed, on the same stream, by an asynchronous copy of that word into pinned host memory and an
    # event.  check() looks at the copies whose event has completed (never blocks); finish() synchronises first.  execute()
    # calls check() on entry, so an asynchronous caller learns of a time-out at its next call at the latest, and the
    # waiting path / Stream users call finish().
    def _post_error_word(self, stream):
        if self._mailbox is None:
            from .hip import ErrorMailbox
            self._mailbox = ErrorMailbox()
        self._mailbox.post(self._context.pointer_of(self._counters) + 4, stream, self._strategy[0])

    @on_plan_device
    def check(self):
        """Raise if a completed asynchronous execute() reported invalid results (non-blocking)."""
        if self._mailbox is not None:
            self._handle_errors(self._mailbox.collect(False))

    @on_plan_device
    def finish(self):
        """Wait for the plan's stream, then raise if any execute() since the last check reported invalid results."""
        self._context.wait()
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
            if strategy == "fused2x" and (word & 4):
                raise RuntimeError("pyfft_amd: the XCD-local launch left an XCD without work-groups (results invalid)")
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
        if self._mailbox is not None:
            self.check()
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

        ctx.createQueue(args)
        ctx.order_scratch()
        self._enqueue(batch, is_inplace, inverse, bufs0, bufs1)

        # global wait setting has lower priority than the local one (plan.py:250-253)
        wait = self._wait_for_finish
        if wait_for_finish is not None:
            wait = wait_for_finish

        if wait:
            self.finish()
        else:
            ctx.flush()
            return ctx.getQueue()

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
        streams and events) instead of at garbage collection.  The plan stays usable: everything is re-created on demand."""
        try:
            self.finish()
        finally:
            self._tempmemobj = None
            self._counters = None
            self._xcd2_scratch = None
            self._side_streams = None
            self._side_events = None
            self._mailbox = None
            self._last_batch_size = 0
            self._last_call_key = None

    # ------------------------------------------------------------------------------------
    # introspection helpers used by bench.py / tests (not part of the reference API)
    def pass_list(self):
        return list(self._kernels)

    @on_plan_device
    def strategy(self, batch):
        self._prepare(int(batch))
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

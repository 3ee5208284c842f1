"""GenericFFTPlan: the two extensions the reference lists as TODO and never had (TODO.txt:6-8, doc/source/index.rst:231):

  * "2D/3D tiled batch support (... to transform several tiles of big 2D/3D array in one pass)":
        Plan(tile_shape, parent_shape=parent)   transforms every non-overlapping tile of `parent`
  * "support for non-power-of-2 sized arrays":
        Plan(shape, any_size=True)              any axis length >= 1 (Bluestein's algorithm per such axis)

Both are OPT-IN keyword arguments: without them `Plan()` behaves exactly like the reference (a size that is not a power of
two is a ValueError, plan.py:23-24).  Neither is a hot path: they are built from the power-of-two plans (plan.py) plus two
streaming helper kernels of the C ABI (`mifft_aux_copy`, `mifft_aux_mul_rows`):

    gather   user buffers (interleaved or split planes, dense or tiles of a parent array) -> dense interleaved work array
    per axis  * power of two: the lines of the axis gathered into dense rows, one batched ROW plan, scattered back
                (the contiguous axis runs in place on the work array)
              * a SMOOTH length n = 2^a 3^b 5^c 7^d up to 4096 (fp32) / 2048 (fp64): the lines gathered into dense rows, ONE
                mixed-radix launch (csrc/fft_mixed.hip: radix-3 / 5 / 7 butterflies next to the power-of-two ones), scattered back
              * any other length n: Bluestein -- rows a[j] = x[j] * c[j] zero-padded to m = 2^k >= 2n - 1 with the chirp
                c[j] = exp(-i pi j^2 / n), A = FFT_m(a), A *= FFT_m(b) (b = conj chirp, wrapped, evaluated on the host in
                float64; its spectrum is computed ONCE, when the plan is built, by this engine's own float64 kernels on the
                device: _device_fft), y = IFFT_m(A), X[k] = y[k] * c[k]
    scatter  work array -> user output, with the plan's scale rule (kernel.py:23-37) and the conjugation trick for the
             inverse transform

Cost: two streaming passes around every axis (six to ten HBM round trips per non-power-of-two axis); results match
the host FFT of the complex128-upcast input within the tolerances stated in tests/test_generic_gpu.py.
"""

import ctypes

import numpy

from . import _debug as D
from . import _native as N
from .plan import FFTPlan, normalize_shape, on_plan_device, _twiddle_table


def _is_pow2(n):
    return n >= 1 and (n & (n - 1)) == 0


def _chirp(n, complex_dtype):
    """c[j] = exp(-i*pi*j^2/n), j < n, with the phase reduced exactly (j^2 mod 2n) and evaluated in float64"""
    j = numpy.arange(n, dtype=numpy.int64)
    ph = (j * j) % (2 * n)
    ang = -numpy.pi * ph.astype(numpy.float64) / float(n)
    return (numpy.cos(ang) + 1j * numpy.sin(ang))


class _Axis(object):
    __slots__ = ("n", "m", "plan", "chirp", "bhat", "pow2", "mixed_tw", "blue")


class _SubContext(object):
    """What the inner power-of-two plans see: the outer plan's context with the stream choice frozen (the outer execute()
    has already picked the stream of this call; an inner plan must not pick another one)."""

    _guard = False      # the outer plan has made its device current before an inner plan runs

    def __init__(self, ctx):
        self._ctx = ctx
        self.compute_units = ctx.compute_units
        self.machine = ctx.machine
        self.allocate = ctx.allocate
        self.allocate_raw = ctx.allocate_raw
        self.upload = ctx.upload
        self.pointer_of = ctx.pointer_of

    def createQueue(self, buffers=()):
        pass

    def order_scratch(self, capturing=None):
        pass            # (the outer execute() has ordered the stream of this call behind the previous one)

    def stream_handle(self):
        return self._ctx.stream_handle()

    def capturing(self):
        return self._ctx.capturing()

    def wait(self):
        self._ctx.wait()

    def flush(self):
        pass

    def getQueue(self):
        return self._ctx.getQueue()


class GenericFFTPlan(object):
    """Same call interface as FFTPlan (execute bound by layout, normalize / scale, wait_for_finish, batch)."""

    def __init__(self, context, shape, dtype=numpy.complex64, normalize=True, wait_for_finish=None, fast_math=True,
                 scale=1.0, parent_shape=None, any_size=False):
        self._dim, xyz = normalize_shape(shape)
        for v in xyz:
            if not isinstance(v, (int, numpy.integer)) or isinstance(v, bool) or v < 1:
                raise ValueError("Wrong shape")
        self._xyz = tuple(int(v) for v in xyz)
        if not any_size and not all(_is_pow2(v) for v in self._xyz):
            raise ValueError("Array dimensions must be powers of two")
        self._size = self._xyz[0] * self._xyz[1] * self._xyz[2]
        if self._size < 2:
            raise ValueError("Array must have at least two elements")
        try:
            dt = numpy.dtype(dtype)
        except TypeError:
            raise ValueError("Data type " + str(dtype) + " is not supported")
        if dt in (numpy.dtype(numpy.complex64), numpy.dtype(numpy.float32)):
            self._precision, self._cdtype = N.F32, numpy.dtype(numpy.complex64)
        elif dt in (numpy.dtype(numpy.complex128), numpy.dtype(numpy.float64)):
            self._precision, self._cdtype = N.F64, numpy.dtype(numpy.complex128)
        else:
            raise ValueError("Data type " + str(dtype) + " is not supported")
        self._split = dt.kind == "f"
        self._context = context
        self._normalize = normalize
        self._scale = float(scale)
        self._wait_for_finish = wait_for_finish

        # tiles of a parent array: parent dims (x, y, z) must be multiples of the tile's
        if parent_shape is not None:
            pdim, pxyz = normalize_shape(parent_shape)
            if pdim != self._dim:
                raise ValueError("parent_shape must have as many axes as the tile shape")
            self._parent = tuple(int(v) for v in pxyz)
            for p, t in zip(self._parent, self._xyz):
                if p < t or p % t:
                    raise ValueError("every parent axis must be a multiple of the tile axis")
            self._counts = tuple(p // t for p, t in zip(self._parent, self._xyz))
        else:
            self._parent = self._xyz
            self._counts = (1, 1, 1)
        self._ntiles = self._counts[0] * self._counts[1] * self._counts[2]

        # Which path runs is decided FIRST; the inner power-of-two plans (one batched 1-D plan per distinct row length, or one N-D
        # plan over all tiles; they share this plan's context = stream) are built only for the work-array path, at the end
        self._sub = _SubContext(context)
        self._rowplans = {}
        self._ndplan = None
        self._shape_arg = shape
        self._tables = []
        all_pow2 = all(_is_pow2(v) for v in self._xyz)
        # tiles of a parent array whose shape has such a kernel: ONE launch straight on the parent array (csrc/fft_nd2t.hpp): no
        # gather, no scatter, no work array -- only the three w(len)^k tables of a MIFFT_PASS_ND pass
        self._tiled = False
        self._tiled_tables = None
        tx, ty, tz = self._xyz
        if all_pow2 and parent_shape is not None and (tx > 1) + (ty > 1) + (tz > 1) >= 2 and not D.no_tiled_kernel() and \
                N.lib.mifft_nd_shape_supported(self._precision, tx, ty, tz, N.VARIANT_INTERLEAVED_ONLY) == 0 and \
                N.lib.mifft_nd_tiled_supported(self._precision, tx, ty, tz) == 0:
            self._tiled = True
            self._tiled_tables = tuple(self._upload(_twiddle_table(n, n, 1, self._cdtype)) if n > 1 else None for n in self._xyz)
        self._axes = []
        self._direct_long = None
        for n in self._xyz:
            ax = _Axis()
            ax.n = n
            ax.pow2 = _is_pow2(n)
            ax.mixed_tw = None
            ax.blue = None
            if not ax.pow2 and N.lib.mifft_mixed_supported(self._precision, n) == 0:
                # smooth length (2^a 3^b 5^c 7^d): ONE mixed-radix launch on the dense rows instead of Bluestein's three transforms
                k = numpy.arange(n, dtype=numpy.float64)
                ang = -2.0 * numpy.pi * k / float(n)
                ax.mixed_tw = self._upload((numpy.cos(ang) + 1j * numpy.sin(ang)).astype(self._cdtype))
                ax.m, ax.plan, ax.chirp, ax.bhat = n, None, None, None
                self._axes.append(ax)
                continue
            if not ax.pow2 and self._xyz[1] == 1 and self._xyz[2] == 1 and not self._split and self._ntiles == 1:
                # 1-D smooth length beyond one tile: n = n1 * n2 in TWO mixed-radix launches (lines of n1 stored as rows and
                # twiddled, then lines of n2: csrc/fft_mixed.hip, mifft_launch_mixed_long) instead of Bluestein
                n1, n2 = ctypes.c_int32(0), ctypes.c_int32(0)
                if N.lib.mifft_mixed_long_split(self._precision, n, ctypes.byref(n1), ctypes.byref(n2)) == 0:
                    self._direct_long = self._long_tables(n, n1.value, n2.value)
                    ax.m, ax.plan, ax.chirp, ax.bhat = n, None, None, None
                    self._axes.append(ax)
                    continue
            mb = ctypes.c_int32(0)
            if not ax.pow2 and N.lib.mifft_bluestein_padded(self._precision, n, ctypes.byref(mb)) == 0:
                # any other length whose padded rows fit a tile: Bluestein's algorithm in ONE launch (both m-point transforms
                # inside LDS, csrc/fft_mixed.hip) instead of three batched transforms and four streaming copies
                m = mb.value
                c = _chirp(n, self._cdtype)
                b = numpy.zeros(m, numpy.complex128)
                b[:n] = numpy.conj(c)
                b[m - n + 1:] = numpy.conj(c[1:][::-1])
                k = numpy.arange(m, dtype=numpy.float64)
                ang = -2.0 * numpy.pi * k / float(m)
                ax.blue = (m, self._upload((numpy.cos(ang) + 1j * numpy.sin(ang)).astype(self._cdtype)),
                           self._upload(c.astype(self._cdtype)), self._upload((self._device_fft(b) / m).astype(self._cdtype)))
                ax.m, ax.plan, ax.chirp, ax.bhat = n, None, None, None
                self._axes.append(ax)
                continue
            ax.m = n if ax.pow2 else 1 << int(numpy.ceil(numpy.log2(2 * n - 1)))
            ax.plan = None                    # a batched ROW plan of ax.m points, built below if the work-array path runs
            ax.chirp = ax.bhat = None
            if not ax.pow2 and not self._tiled:
                c = _chirp(n, self._cdtype)
                b = numpy.zeros(ax.m, numpy.complex128)
                b[:n] = numpy.conj(c)
                b[ax.m - n + 1:] = numpy.conj(c[1:][::-1])
                ax.chirp = self._upload(c.astype(self._cdtype))
                ax.bhat = self._upload(self._device_fft(b).astype(self._cdtype))
            self._axes.append(ax)
        # 1-D smooth length, interleaved, dense: the mixed-radix launch reads the user's input and writes the user's output itself
        # (conjugation of the inverse direction and the scale included): one HBM round trip, no work array
        self._direct_mixed = (self._xyz[1] == 1 and self._xyz[2] == 1 and self._axes[0].mixed_tw is not None
                              and not self._split and self._ntiles == 1)
        # ... and a 1-D length with the one-launch Bluestein kernel likewise
        self._direct_blue = (self._xyz[1] == 1 and self._xyz[2] == 1 and self._axes[0].blue is not None
                             and not self._split and self._ntiles == 1)
        # N-D, every axis a smooth length the mixed-radix kernel takes (powers of two included), at least one of them not a power of
        # two, interleaved, dense: ONE launch per axis straight on the user's buffers -- the x axis as rows from input to output,
        # the slower axes as lines of the output array in place (mifft_launch_mixed_lines); no gather, no scatter, no work array
        self._direct_nd = None
        self._direct_nd1 = False
        self._direct_nd_planes = False
        if not self._direct_mixed and not self._split and self._ntiles == 1 and any(ax.mixed_tw is not None for ax in self._axes) and \
                all(ax.n == 1 or N.lib.mifft_mixed_supported(self._precision, ax.n) == 0 for ax in self._axes):
            tabs = []
            for ax in self._axes:
                if ax.n == 1:
                    tabs.append(None)
                elif ax.mixed_tw is not None:
                    tabs.append(ax.mixed_tw)
                else:
                    k = numpy.arange(ax.n, dtype=numpy.float64)
                    ang = -2.0 * numpy.pi * k / float(ax.n)
                    tabs.append(self._upload((numpy.cos(ang) + 1j * numpy.sin(ang)).astype(self._cdtype)))
            self._direct_nd = tabs
            # ... and the whole transform in ONE launch when it fits a work-group's LDS (csrc/fft_mixed_nd.hip, round 4): (100, 100)
            # is 80 KB; one HBM round trip instead of one per axis
            x, y, z = self._xyz
            if N.lib.mifft_mixed_nd_supported(self._precision, x, y, z) == 0 and not D.no_mixed_nd():
                self._direct_nd1 = True
            elif z > 1 and x > 1 and y > 1 and N.lib.mifft_mixed_nd_supported(self._precision, x, y, 1) == 0 and not D.no_mixed_nd():
                # a 3-D shape beyond one tile whose (y, x) PLANES fit one: the planes in one launch (they are just more transforms),
                # then the z lines -- two HBM round trips instead of three ((60, 60, 60): 216000 points)
                self._direct_nd_planes = True
        self._uses_work = not (self._tiled or self._direct_mixed or self._direct_blue or self._direct_nd is not None
                               or self._direct_long is not None)
        if self._uses_work:
            if all_pow2:
                # every axis a power of two (tiles of a parent array): ONE N-D plan on the dense work array
                self._ndplan = FFTPlan(self._sub, shape, dtype=self._cdtype, normalize=True, wait_for_finish=False)
            else:
                for ax in self._axes:
                    if ax.m > 1 and ax.mixed_tw is None and ax.blue is None:
                        ax.plan = self._rowplan(ax.m)
        self._work = None
        self._rows = None
        self._last_batch = 0
        self._captured = False
        self._capture_keepalive = []
        if self._split:
            self.execute = self._executeSplit
        else:
            self.execute = self._executeInterleaved

    # ------------------------------------------------------------------------------------------------
    def _rowplan(self, m):
        if m not in self._rowplans:
            self._rowplans[m] = FFTPlan(self._sub, (m,), dtype=self._cdtype, normalize=True, wait_for_finish=False)
        return self._rowplans[m]

    def _device_fft(self, b):
        """FFT_m of the host vector `b` (complex128) computed ON THE DEVICE by this engine's own float64 kernels, returned as a
        host complex128 array: the spectrum of Bluestein's wrapped conjugate chirp, a plan-build-time constant (an FFT engine
        transforms its own chirp; the reference lists non-power-of-two sizes as a TODO, TODO.txt:8).  Always float64, also for
        fp32 plans, so that the table is rounded ONCE to the working precision like every other table of the engine
        (plan._twiddle_table).  m a power of two: a dense FFTPlan of m points; m smooth: the mixed-radix row kernel, or the
        two-launch long form beyond its tile (include/mifft.h: mifft_launch_mixed_rows / mifft_launch_mixed_long)."""
        ctx = self._context
        cd = numpy.dtype(numpy.complex128)
        host = numpy.ascontiguousarray(b, dtype=cd)
        m = int(host.shape[0])
        stream = ctx.stream_handle()
        keep = []

        def dev(arr):
            arr = numpy.ascontiguousarray(arr)
            mem = ctx.allocate_raw(arr.nbytes)
            ctx.upload(mem, arr)
            keep.append(mem)
            return ctx.pointer_of(mem)

        def roots(count, step, period):
            k = numpy.arange(count, dtype=numpy.float64) * float(step)
            ang = -2.0 * numpy.pi * numpy.fmod(k, float(period)) / float(period)
            return dev((numpy.cos(ang) + 1j * numpy.sin(ang)).astype(cd))

        src = dev(host)
        dst = src
        if _is_pow2(m):
            plan = FFTPlan(self._sub, (m,), dtype=cd, normalize=False, wait_for_finish=True)
            plan.execute(src)
            plan.close()
        elif N.lib.mifft_mixed_supported(N.F64, m) == 0:
            N.check(N.lib.mifft_launch_mixed_rows(N.F64, m, 1, m, m, src, src, roots(m, 1, m), 0, 1.0, stream), "mifft_launch_mixed_rows")
        else:
            n1, n2 = ctypes.c_int32(0), ctypes.c_int32(0)
            N.check(N.lib.mifft_mixed_long_split(N.F64, m, ctypes.byref(n1), ctypes.byref(n2)), "mifft_mixed_long_split(%d)" % m)
            shift = max(1, (int(m - 1).bit_length() + 1) // 2)
            dst = dev(numpy.zeros(m, cd))
            N.check(N.lib.mifft_launch_mixed_long(N.F64, n1.value, n2.value, 1, src, dst, dst, roots(n1.value, 1, n1.value),
                                                  roots(n2.value, 1, n2.value), roots(1 << shift, 1, m),
                                                  roots(((m - 1) >> shift) + 1, 1 << shift, m), shift, 0, 1.0, stream),
                    "mifft_launch_mixed_long")
        out = numpy.empty(m, cd)
        N.check(N.lib.mifft_memcpy_d2h(out.ctypes.data, dst, out.nbytes, stream), "mifft_memcpy_d2h")
        N.check(N.lib.mifft_stream_sync(stream), "mifft_stream_sync")
        del keep[:]
        return out

    def _upload(self, host):
        host = numpy.ascontiguousarray(host)
        mem = self._context.allocate_raw(host.nbytes)
        self._context.upload(mem, host)
        self._tables.append(mem)
        return self._context.pointer_of(mem)

    def _copy(self, dims, sstride, dstride, src0, src1, dst0, dst1, src_valid0=0, conj_in=False, conj_out=False, mult=None,
              scale=1.0, src_split=False, dst_split=False):
        c = N.MifftCopy()
        c.precision = self._precision
        c.ndim = len(dims)
        for i, (d, s, t) in enumerate(zip(dims, sstride, dstride)):
            c.dims[i], c.src_stride[i], c.dst_stride[i] = int(d), int(s), int(t)
        c.src_valid0 = int(src_valid0)
        c.src_split, c.dst_split = int(bool(src_split)), int(bool(dst_split))
        c.conj_in, c.conj_out = int(bool(conj_in)), int(bool(conj_out))
        c.mult = mult
        c.scale = float(scale)
        N.check(N.lib.mifft_aux_copy(ctypes.byref(c), src0, src1, dst0, dst1, self._context.stream_handle()), "mifft_aux_copy")

    def _user_dims_strides(self, items):
        """index space (x, y, z, cx, cy, item*cz) of `items` parent arrays and its strides in the user's buffers / the dense
        work array [item][cz][cy][cx][z][y][x]"""
        tx, ty, tz = self._xyz
        px, py, pz = self._parent
        cx, cy, cz = self._counts
        dims = (tx, ty, tz, cx, cy, items * cz)
        user = (1, px, px * py, tx, ty * px, tz * px * py)
        tile = tx * ty * tz
        work = (1, tx, tx * ty, tile, tile * cx, tile * cx * cy)
        return dims, user, work

    def _long_tables(self, n, n1, n2):
        """(n1, n2, w(n1)^m, w(n2)^m, lo, hi, shift) for mifft_launch_mixed_long: w(n)^e = lo[e & (2^shift - 1)] * hi[e >> shift],
        every entry evaluated in float64."""
        def roots(count, step, period):
            k = numpy.arange(count, dtype=numpy.float64) * float(step)
            ang = -2.0 * numpy.pi * numpy.fmod(k, float(period)) / float(period)
            return self._upload((numpy.cos(ang) + 1j * numpy.sin(ang)).astype(self._cdtype))
        shift = max(1, (int(n - 1).bit_length() + 1) // 2)
        lo = roots(1 << shift, 1, n)
        hi = roots(((n - 1) >> shift) + 1, 1 << shift, n)
        return (n1, n2, roots(n1, 1, n1), roots(n2, 1, n2), lo, hi, shift)

    def _epilogue(self, wait_for_finish):
        """The end of every execute (plan.py:250-259): wait (and check for errors) or hand the stream back."""
        wait = self._wait_for_finish if wait_for_finish is None else wait_for_finish
        if wait:
            self.finish()
            return None
        self._context.flush()
        return self._context.getQueue()

    def _prepare(self, batch):
        if batch == self._last_batch:
            return
        if self._captured and (self._work is not None or self._rows is not None):
            self._capture_keepalive.append((self._work, self._rows))      # (a recorded graph replays on them: hip.Graph docstring)
        self._last_batch = batch
        if not self._uses_work:
            self._work = None           # no work arrays (a long smooth transform in place allocates its scratch on demand)
            return
        isz = self._cdtype.itemsize
        nt = batch * self._ntiles
        self._work = self._context.allocate(nt * self._size * isz)
        worst = max((self._size // ax.n) * ax.m for ax in self._axes)
        self._rows = self._context.allocate(nt * worst * isz)

    @on_plan_device
    def _execute(self, wait_for_finish, inverse, batch, ins, outs):
        ctx = self._context
        batch = int(batch)
        if batch < 1:
            raise ValueError("batch must be positive")
        self.check()
        self._prepare(batch)
        ptr = ctx.pointer_of
        ctx.createQueue(ins + outs)
        ctx.order_scratch()
        if ctx.capturing():
            # recorded into a graph (work arrays must exist: one eager execute of the batch first): the graph keeps this plan alive,
            # the plan keeps the work arrays of the recorded batch
            from .hip import Graph
            self._captured = True
            Graph.retain(self)
        if self._tiled:
            return self._execute_tiled(wait_for_finish, bool(inverse), batch, [ptr(b) for b in ins], [ptr(b) for b in outs])
        if self._direct_nd1:
            inv = bool(inverse)
            factor = self._scale if not inv else 1.0 / ((self._size if self._normalize else 1.0) * self._scale)
            x, y, z = self._xyz
            twx, twy, twz = self._direct_nd
            N.check(N.lib.mifft_launch_mixed_nd(self._precision, x, y, z, batch, ptr(ins[0]), ptr(outs[0]), twx, twy, twz, 1 if inv else 0,
                                                factor, ctx.stream_handle()), "mifft_launch_mixed_nd")
            return self._epilogue(wait_for_finish)
        if self._direct_nd_planes:
            inv = bool(inverse)
            factor = self._scale if not inv else 1.0 / ((self._size if self._normalize else 1.0) * self._scale)
            x, y, z = self._xyz
            twx, twy, twz = self._direct_nd
            N.check(N.lib.mifft_launch_mixed_nd(self._precision, x, y, 1, batch * z, ptr(ins[0]), ptr(outs[0]), twx, twy, None, 2 if inv else 0,
                                                1.0, ctx.stream_handle()), "mifft_launch_mixed_nd")
            N.check(N.lib.mifft_launch_mixed_lines(self._precision, z, batch, x * y, ptr(outs[0]), ptr(outs[0]), twz, 0, 1 if inv else 0,
                                                   factor, ctx.stream_handle()), "mifft_launch_mixed_lines")
            return self._epilogue(wait_for_finish)
        if self._direct_nd is not None:
            inv = bool(inverse)
            factor = self._scale if not inv else 1.0 / ((self._size if self._normalize else 1.0) * self._scale)
            todo = [a for a, ax in enumerate(self._axes) if ax.n > 1]
            src, inner = ptr(ins[0]), 1
            for a, ax in enumerate(self._axes):            # x, y, z
                if ax.n > 1:
                    first, last = a == todo[0], a == todo[-1]
                    outer = batch * self._size // (ax.n * inner)
                    N.check(N.lib.mifft_launch_mixed_lines(self._precision, ax.n, outer, inner, src, ptr(outs[0]), self._direct_nd[a],
                                                           1 if (inv and first) else 0, 1 if (inv and last) else 0,
                                                           factor if last else 1.0, ctx.stream_handle()), "mifft_launch_mixed_lines")
                    src = ptr(outs[0])
                inner *= ax.n
            return self._epilogue(wait_for_finish)
        if self._direct_long is not None:
            n1, n2, tw1, tw2, lo, hi, shift = self._direct_long
            n = n1 * n2
            inv = bool(inverse)
            factor = self._scale if not inv else 1.0 / ((n if self._normalize else 1.0) * self._scale)
            src, dst = ptr(ins[0]), ptr(outs[0])
            mid = dst
            if src == dst:                  # in place: the transposing first pass needs somewhere else to write
                if self._work is None:
                    self._work = ctx.allocate(batch * n * self._cdtype.itemsize)
                mid = ptr(self._work)
            N.check(N.lib.mifft_launch_mixed_long(self._precision, n1, n2, batch, src, mid, dst, tw1, tw2, lo, hi, shift,
                                                  1 if inv else 0, factor, ctx.stream_handle()), "mifft_launch_mixed_long")
            return self._epilogue(wait_for_finish)
        if self._direct_blue:
            n = self._xyz[0]
            m, tw, chirp, bhat = self._axes[0].blue
            inv = bool(inverse)
            factor = self._scale if not inv else 1.0 / ((n if self._normalize else 1.0) * self._scale)
            N.check(N.lib.mifft_launch_bluestein_rows(self._precision, n, m, batch, n, n, ptr(ins[0]), ptr(outs[0]), tw, chirp, bhat,
                                                      1 if inv else 0, factor, ctx.stream_handle()), "mifft_launch_bluestein_rows")
            return self._epilogue(wait_for_finish)
        if self._direct_mixed:
            n = self._xyz[0]
            inv = bool(inverse)
            factor = self._scale if not inv else 1.0 / ((n if self._normalize else 1.0) * self._scale)
            N.check(N.lib.mifft_launch_mixed_rows(self._precision, n, batch, n, n, ptr(ins[0]), ptr(outs[0]), self._axes[0].mixed_tw,
                                                  1 if inv else 0, factor, ctx.stream_handle()), "mifft_launch_mixed_rows")
            return self._epilogue(wait_for_finish)
        nt = batch * self._ntiles
        work, rows = ptr(self._work), ptr(self._rows)
        in0, in1 = ptr(ins[0]), (ptr(ins[1]) if self._split else None)
        out0, out1 = ptr(outs[0]), (ptr(outs[1]) if self._split else None)
        inverse = bool(inverse)

        # the transform is forward on conj(input), conjugated at the end, for the inverse direction
        dims, user, dense = self._user_dims_strides(batch)
        self._copy(dims, user, dense, in0, in1, work, None, conj_in=inverse, src_split=self._split)

        inner = 1
        if self._ndplan is not None:
            self._ndplan.execute(work, batch=nt, wait_for_finish=False)
        for a, ax in enumerate(self._axes):           # x, y, z
            n, m = ax.n, ax.m
            if n > 1 and self._ndplan is None:
                outer = nt * self._size // (n * inner)
                lines = (n, inner, outer)              # index space of the axis' lines in the work array
                wstride = (inner, 1, n * inner)
                rstride = (1, m, m * inner)            # dense rows [outer][inner][m]
                if ax.mixed_tw is not None:
                    # (the conjugation of the inverse direction and the scale live in the gather / scatter: always forward, scale 1)
                    if inner == 1:
                        N.check(N.lib.mifft_launch_mixed_rows(self._precision, n, outer, n, n, work, work, ax.mixed_tw, 0, 1.0,
                                                              ctx.stream_handle()), "mifft_launch_mixed_rows")
                    else:
                        self._copy(lines, wstride, rstride, work, None, rows, None)
                        N.check(N.lib.mifft_launch_mixed_rows(self._precision, n, outer * inner, n, n, rows, rows, ax.mixed_tw, 0, 1.0,
                                                              ctx.stream_handle()), "mifft_launch_mixed_rows")
                        self._copy(lines, rstride, wstride, rows, None, work, None)
                elif ax.blue is not None:
                    bm, btw, bchirp, bbhat = ax.blue
                    if inner == 1:
                        N.check(N.lib.mifft_launch_bluestein_rows(self._precision, n, bm, outer, n, n, work, work, btw, bchirp, bbhat, 0, 1.0,
                                                                  ctx.stream_handle()), "mifft_launch_bluestein_rows")
                    else:
                        self._copy(lines, wstride, rstride, work, None, rows, None)
                        N.check(N.lib.mifft_launch_bluestein_rows(self._precision, n, bm, outer * inner, n, n, rows, rows, btw, bchirp, bbhat,
                                                                  0, 1.0, ctx.stream_handle()), "mifft_launch_bluestein_rows")
                        self._copy(lines, rstride, wstride, rows, None, work, None)
                elif ax.pow2 and inner == 1:
                    ax.plan.execute(work, batch=outer, wait_for_finish=False)
                elif ax.pow2:
                    self._copy(lines, wstride, rstride, work, None, rows, None)
                    ax.plan.execute(rows, batch=outer * inner, wait_for_finish=False)
                    self._copy(lines, rstride, wstride, rows, None, work, None)
                else:
                    self._copy((m, inner, outer), wstride, rstride, work, None, rows, None, src_valid0=n, mult=ax.chirp)
                    ax.plan.execute(rows, batch=outer * inner, wait_for_finish=False)
                    N.check(N.lib.mifft_aux_mul_rows(self._precision, rows, ax.bhat, outer * inner, m, ctx.stream_handle()),
                            "mifft_aux_mul_rows")
                    ax.plan.execute(rows, batch=outer * inner, inverse=True, wait_for_finish=False)   # normalised: 1/m
                    self._copy(lines, rstride, wstride, rows, None, work, None, mult=ax.chirp)
            inner *= n

        # the scale rule of _FFTKernel.getScaleCoeffFunc (kernel.py:23-37)
        if not inverse:
            factor = self._scale
        else:
            factor = 1.0 / ((self._size if self._normalize else 1.0) * self._scale)
        self._copy(dims, dense, user, work, None, out0, out1, conj_out=inverse, scale=factor, dst_split=self._split)

        return self._epilogue(wait_for_finish)

    def _execute_tiled(self, wait_for_finish, inverse, batch, src, dst):
        """Every tile transformed where it lies: one MIFFT_PASS_ND launch over batch * tiles transforms with the parent's pitches."""
        ctx = self._context
        tx, ty, tz = self._xyz
        px, py, pz = self._parent
        cx, cy, cz = self._counts
        d = N.MifftPass()
        d.kind, d.precision, d.layout, d.inverse = N.PASS_ND, self._precision, (N.SPLIT if self._split else N.INTERLEAVED), 1 if inverse else 0
        d.L, d.M, d.S = tx, ty, tz
        d.outer = batch * self._ntiles
        d.outer_stride_in = d.outer_stride_out = self._size
        d.scale = self._scale if not inverse else 1.0 / ((self._size if self._normalize else 1.0) * self._scale)
        d.tw_L, d.tw_lo, d.tw_hi = self._tiled_tables
        t = N.MifftTiling()
        t.pitch_y, t.pitch_z, t.parent_elems = px, px * py, px * py * pz
        t.cx, t.cy, t.cz = cx, cy, cz
        if self._split:   # re / im planes of the parent, same launch (round 4)
            N.check(N.lib.mifft_launch_nd_tiled_split(ctypes.byref(d), ctypes.byref(t), src[0], src[1], dst[0], dst[1], ctx.stream_handle()),
                    "mifft_launch_nd_tiled_split")
        else:
            N.check(N.lib.mifft_launch_nd_tiled(ctypes.byref(d), ctypes.byref(t), src[0], dst[0], ctx.stream_handle()), "mifft_launch_nd_tiled")
        return self._epilogue(wait_for_finish)

    def _inner_plans(self):
        plans = [self._ndplan] if self._ndplan is not None else []
        plans += [ax.plan for ax in self._axes if getattr(ax, "plan", None) is not None]
        return plans

    @on_plan_device
    def finish(self):
        """Wait for the stream, then drain every inner plan's error mailbox (an inner power-of-two plan that runs a persistent
        kernel posts its dependency time-outs to ITS mailbox; they are this plan's errors)."""
        self._context.wait()
        for p in self._inner_plans():
            p.finish()

    @on_plan_device
    def check(self):
        """Non-blocking: raise if a completed asynchronous execute() of an inner plan reported invalid results."""
        for p in self._inner_plans():
            p.check()

    def _executeInterleaved(self, data_in, data_out=None, inverse=False, batch=1, wait_for_finish=None):
        if data_out is None:
            data_out = data_in
        return self._execute(wait_for_finish, inverse, batch, [data_in], [data_out])

    def _executeSplit(self, data_in_re, data_in_im, data_out_re=None, data_out_im=None, inverse=False, batch=1,
                      wait_for_finish=None):
        if data_out_re is None and data_out_im is None:
            data_out_re, data_out_im = data_in_re, data_in_im
        elif data_out_re is None or data_out_im is None:
            raise ValueError("both output planes must be given")
        return self._execute(wait_for_finish, inverse, batch, [data_in_re, data_in_im], [data_out_re, data_out_im])

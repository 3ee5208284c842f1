"""HIP backend of the pyfft API for MI355X: `Plan()` factory and execution context.

Counterpart of the reference's pyfft/cuda.py.  Usage is the reference's with the module name
changed:

    from pyfft_amd.hip import Plan          # was: from pyfft.cuda import Plan
    plan = Plan((1024, 1024), dtype=numpy.complex64, stream=my_stream)
    plan.execute(gpu_buf)                   # in place
    plan.execute(gpu_in, gpu_out, inverse=True, batch=8)

Buffers may be anything that exposes a device address: a `DeviceArray` from this module, a
torch-ROCm tensor (`data_ptr()`), an object with `gpudata` / `ptr` / `__cuda_array_interface__`,
a ctypes pointer or a plain int ("GPUArray or anything that can be cast to a pointer",
doc/source/index.rst:243-246, cuda.py:36-39).
"""

import ctypes
import gc
import threading

import numpy

from . import _native as N
from .plan import FFTPlan
from .machine import Machine


def device_pointer(obj):
    """Device address of a buffer-like object (counterpart of the GPUArray -> gpudata unwrapping
    in Function.__call__, cuda.py:36-39)."""
    t = type(obj)
    if t is DeviceArray or t is DeviceAllocation:      # (the common case first: execute() is launch-bound for small plans)
        return obj.ptr
    if obj is None:
        return None
    if isinstance(obj, (int, numpy.integer)) and not isinstance(obj, bool):
        return int(obj)
    if isinstance(obj, DeviceArray) or isinstance(obj, DeviceAllocation):
        return obj.ptr
    if hasattr(obj, "data_ptr"):            # torch tensor
        return int(obj.data_ptr())
    if hasattr(obj, "gpudata"):             # PyCUDA-style
        return int(obj.gpudata)
    if hasattr(obj, "__cuda_array_interface__"):
        return int(obj.__cuda_array_interface__["data"][0])
    if hasattr(obj, "__hip_array_interface__"):
        return int(obj.__hip_array_interface__["data"][0])
    if isinstance(obj, ctypes.c_void_p):
        return obj.value
    if hasattr(obj, "ptr"):
        return int(obj.ptr)
    try:
        return int(obj)
    except (TypeError, ValueError):
        raise TypeError("cannot obtain a device pointer from %r" % (type(obj),))


class DeviceAllocation(object):
    """Owning handle of a hipMalloc block (counterpart of pycuda.driver.DeviceAllocation)."""

    def __init__(self, nbytes):
        p = ctypes.c_void_p()
        N.check(N.lib.mifft_malloc(ctypes.byref(p), int(nbytes)), "mifft_malloc(%d)" % nbytes)
        self.ptr = p.value
        self.nbytes = int(nbytes)

    def free(self):
        if getattr(self, "ptr", None):
            N.lib.mifft_free(self.ptr)
            self.ptr = None

    def __int__(self):
        return self.ptr

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceArray(object):
    """Minimal device array (counterpart of pycuda.gpuarray.GPUArray as the tests use it:
    allocate, to_gpu, get)."""

    def __init__(self, shape, dtype, allocation=None):
        self.shape = tuple(shape) if isinstance(shape, (tuple, list)) else (int(shape),)
        self.dtype = numpy.dtype(dtype)
        self.size = int(numpy.prod(self.shape)) if len(self.shape) else 1
        self.nbytes = self.size * self.dtype.itemsize
        self._alloc = allocation if allocation is not None else DeviceAllocation(self.nbytes)
        self.ptr = self._alloc.ptr

    @property
    def gpudata(self):
        return self.ptr

    def set(self, host, stream=None):
        host = numpy.ascontiguousarray(host, dtype=self.dtype)
        assert host.nbytes == self.nbytes
        N.check(N.lib.mifft_memcpy_h2d(self.ptr, host.ctypes.data, self.nbytes, _stream_handle(stream)),
                "mifft_memcpy_h2d")
        return self

    def get(self, stream=None):
        host = numpy.empty(self.shape, self.dtype)
        N.check(N.lib.mifft_memcpy_d2h(host.ctypes.data, self.ptr, self.nbytes, _stream_handle(stream)),
                "mifft_memcpy_d2h")
        return host

    def __int__(self):
        return self.ptr


def to_gpu(host):
    host = numpy.ascontiguousarray(host)
    return DeviceArray(host.shape, host.dtype).set(host)


class Stream(object):
    """Owning wrapper of a hipStream_t (counterpart of pycuda.driver.Stream)."""

    def __init__(self):
        h = ctypes.c_void_p()
        N.check(N.lib.mifft_stream_create(ctypes.byref(h)), "mifft_stream_create")
        self.handle = h.value

    def synchronize(self):
        N.check(N.lib.mifft_stream_sync(self.handle), "mifft_stream_sync")

    finish = synchronize  # PyOpenCL spelling (cl.py queue.finish())

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                N.lib.mifft_stream_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


def _stream_handle(stream):
    """hipStream_t value of a stream-like object: Stream, torch.cuda.Stream (cuda_stream), int, None."""
    if stream is None:
        return None
    if isinstance(stream, Stream):
        return stream.handle
    if hasattr(stream, "cuda_stream"):
        return int(stream.cuda_stream)
    if hasattr(stream, "handle"):
        h = stream.handle
        return h.value if isinstance(h, ctypes.c_void_p) else int(h)
    if isinstance(stream, ctypes.c_void_p):
        return stream.value
    return int(stream)


class Event(object):
    def __init__(self):
        h = ctypes.c_void_p()
        N.check(N.lib.mifft_event_create(ctypes.byref(h)), "mifft_event_create")
        self.handle = h.value

    def record(self, stream=None):
        N.check(N.lib.mifft_event_record(self.handle, _stream_handle(stream)), "mifft_event_record")
        return self

    def synchronize(self):
        N.check(N.lib.mifft_event_sync(self.handle), "mifft_event_sync")

    def time_since(self, start):
        ms = ctypes.c_float()
        N.check(N.lib.mifft_event_elapsed_ms(ctypes.byref(ms), start.handle, self.handle), "mifft_event_elapsed_ms")
        return float(ms.value)

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                N.lib.mifft_event_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class Graph(object):
    """Stream capture and hipGraph replay of what plans enqueue (include/mifft.h: mifft_stream_begin_capture ...):

        plan = Plan(shape, stream=s)            # wait_for_finish defaults to False with a stream
        plan.execute(a, b, batch=n)             # once, eagerly: scratch, tables and strategy exist before the capture
        with Graph(s) as g:
            plan.execute(a, b, batch=n)         # recorded, not run
        g.launch(); g.launch(); s.synchronize()

    A captured execute of a persistent strategy runs on a counter set of its own with the memset as a node of the graph, so replays
    and eager executes of the same plan may alternate on one stream.  A torch.cuda.graph() capture around execute() works the
    same way (the plan follows torch's current stream).

    LIFETIME.  A recorded execute bakes RAW DEVICE ADDRESSES into the graph: the caller's buffers and the plan's twiddle tables,
    ring / temp buffer, counter sets and pinned error word.  Two things keep them valid: (i) a Graph holds a reference to every plan
    that executed inside its `with` block, so dropping the plan while the Graph lives frees nothing; (ii) a plan that has been captured
    keeps the scratch of the captured launch alive when a later execute() of another batch, strategy() or close() replaces it
    (FFTPlan._retain_captured_scratch) until the plan is collected or plan.release_captured() is called.  With torch.cuda.graph()
    (or any capture API other than this class) only (ii) applies: keep the plan alive as long as the graph, like the tensors.

    ONE STREAM PER PLAN AT A TIME.  Every captured launch of a plan replays on the SAME third counter set and the same ring / temp
    buffer: launch the graphs recorded from one plan on one stream (replays then queue up behind each other), and do not run an
    eager execute of that plan on ANOTHER stream while a replay is in flight -- the plan orders a later eager execute behind the
    stream the capture was recorded on (Context.order_scratch), not behind a stream a graph was merely launched on.

    Do not drop the last reference to a plan, a stream or a pinned buffer inside the block: releasing them synchronises the device,
    which invalidates an active capture (measured: one run in four of the capture tests failed with "operation not permitted when
    stream is capturing" when the cyclic collector happened to fire inside the window -- hence the collect-and-pause in __enter__;
    the pause is process-wide and reference-counted: the collector comes back when the LAST open capture of any thread ends).
    execute(..., wait_for_finish=True) inside the block is a RuntimeError (waiting synchronises the stream)."""

    _lock = threading.Lock()
    _open = []                 # every Graph between __enter__ and __exit__, of any thread
    _gc_paused = False         # ... while that list is non-empty the cyclic collector is off, if it was on
    _gc_was_enabled = False

    @classmethod
    def _pause_gc(cls, graph):
        with cls._lock:
            if not cls._open:
                cls._gc_was_enabled = gc.isenabled()
                gc.disable()
                cls._gc_paused = True
            cls._open.append(graph)

    @classmethod
    def _resume_gc(cls, graph):
        with cls._lock:
            if graph in cls._open:
                cls._open.remove(graph)
            if not cls._open and cls._gc_paused:
                cls._gc_paused = False
                if cls._gc_was_enabled:
                    gc.enable()

    @classmethod
    def retain(cls, plan):
        """Called by a plan whose execute() is being recorded: every open Graph on the plan's capturing stream keeps the plan alive."""
        try:
            h = plan._context.stream_handle()
        except Exception:
            h = None
        with cls._lock:
            for g in cls._open:
                if g.handle is None and _stream_handle(g.stream) == h and not any(p is plan for p in g._plans):
                    g._plans.append(plan)

    def __init__(self, stream):
        self.stream = stream
        self.handle = None
        self._plans = []           # plans recorded into this graph: alive as long as the graph is

    def __enter__(self):
        # an object that frees device memory when it is collected (a DeviceArray, a plan's scratch) would do so at an arbitrary point of
        # the capture window -- hipFree synchronises the device, which a capturing stream refuses and which invalidates the capture:
        # collect now, and keep the cyclic collector off until the capture has ended (torch.cuda.graph() collects first for the same reason)
        gc.collect()
        Graph._pause_gc(self)
        try:
            N.check(N.lib.mifft_stream_begin_capture(_stream_handle(self.stream)), "mifft_stream_begin_capture")
        except Exception:
            Graph._resume_gc(self)
            raise
        return self

    def __exit__(self, etype, evalue, tb):
        h = ctypes.c_void_p()
        try:
            rc = N.lib.mifft_stream_end_capture(_stream_handle(self.stream), ctypes.byref(h))
        finally:
            Graph._resume_gc(self)
        if etype is None:
            N.check(rc, "mifft_stream_end_capture")
            self.handle = h.value
        elif rc == 0 and h.value:
            N.lib.mifft_graph_destroy(h.value)
        elif rc != 0:
            # the body failed AND the capture could not be ended (it was invalidated): say both, the stream may still be capturing
            raise RuntimeError("pyfft_amd: the capture could not be ended after %r: %s" % (evalue, N.last_error())) from evalue
        return False

    def launch(self, stream=None):
        if not self.handle:
            raise RuntimeError("pyfft_amd: nothing was captured")
        N.check(N.lib.mifft_graph_launch(self.handle, _stream_handle(self.stream if stream is None else stream)), "mifft_graph_launch")

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                N.lib.mifft_graph_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class ErrorMailbox(object):
    """Pinned host words that receive a persistent kernel's error word behind every launch (FFTPlan.check / finish):
    an asynchronous 4-byte copy plus an event on the launch's stream, inspected by the host without synchronising."""

    SLOTS = 64

    def __init__(self):
        p = ctypes.c_void_p()
        N.check(N.lib.mifft_host_alloc(ctypes.byref(p), 4 * self.SLOTS), "mifft_host_alloc")
        self._host = p.value
        self._words = (ctypes.c_uint32 * self.SLOTS).from_address(p.value)
        self._events = [None] * self.SLOTS
        self._pending = []          # (slot, tag) in launch order
        self._stashed = []          # (tag, word) of launches retired by post() when the ring was full
        self._next = 0

    def post(self, dev_ptr, stream, tag):
        if len(self._pending) >= self.SLOTS:
            # the ring is full: retire the oldest launch (wait for it, read its word, keep a non-zero one for the next
            # collect()) BEFORE its slot is reused -- its slot is exactly the one `_next` points at
            slot0, tag0 = self._pending.pop(0)
            self._events[slot0].synchronize()
            word = int(self._words[slot0])
            if word:
                self._stashed.append((tag0, word))
        slot = self._next
        self._next = (slot + 1) % self.SLOTS
        if self._events[slot] is None:
            self._events[slot] = Event()
        self._words[slot] = 0
        N.check(N.lib.mifft_memcpy_d2h_async(self._host + 4 * slot, dev_ptr, 4, _stream_handle(stream)), "mifft_memcpy_d2h_async")
        self._events[slot].record(stream)
        self._pending.append((slot, tag))

    def collect(self, wait):
        """[(tag, word)] of the finished launches whose word is non-zero; wait=True waits for every pending launch."""
        errors, keep = self._stashed, []
        self._stashed = []
        for slot, tag in self._pending:
            ev = self._events[slot]
            if wait:
                ev.synchronize()
            elif keep or N.lib.mifft_event_query(ev.handle) != 0:
                keep.append((slot, tag))       # (launch order: nothing behind an unfinished launch has finished)
                continue
            word = int(self._words[slot])
            if word:
                errors.append((tag, word))
        self._pending = keep
        return errors

    def __del__(self):
        try:
            if getattr(self, "_host", None):
                N.lib.mifft_host_free(self._host)
                self._host = None
        except Exception:
            pass


class ErrorWord(object):
    """A pinned host word the persistent kernels write (system-scope store) when a bounded dependency wait times out: the
    kernel's mifft_fused_sync.error_word.  The host reads it without a copy, an event or a synchronisation -- a non-zero value
    means some launch since the last clear produced invalid results (round 4; rounds 2-3 copied a device word back behind every
    launch, which a 32 MiB execute cannot afford)."""

    def __init__(self):
        p = ctypes.c_void_p()
        N.check(N.lib.mifft_host_alloc(ctypes.byref(p), 64), "mifft_host_alloc")
        self.ptr = p.value
        self._word = ctypes.c_uint32.from_address(p.value)
        self._word.value = 0
        d = ctypes.c_int()
        N.check(N.lib.mifft_get_device(ctypes.byref(d)), "mifft_get_device")
        self._device = d.value           # the device whose kernels write the word (the plan's: created under on_plan_device)

    def take(self):
        """The word's value; clears it when non-zero."""
        v = int(self._word.value)
        if v:
            self._word.value = 0
        return v

    def __del__(self):
        try:
            if getattr(self, "ptr", None):
                # a persistent kernel that is still running may write the word (on a time-out): the device is synchronised before the
                # page goes away (plans are rarely destroyed with work in flight; FFTPlan.close() has synchronised already).  Not while a
                # stream capture is open anywhere in the process, though: a device synchronisation from a finaliser would invalidate the
                # capture (whoever's it is) -- the page then waits in a graveyard that the next finaliser outside a capture empties.
                ptr, dev = self.ptr, self._device
                self.ptr = None
                with Graph._lock:
                    capturing = bool(Graph._open) or _foreign_capture_active()
                    if capturing:
                        ErrorWord._graveyard.append((ptr, dev))
                        return
                    dead = ErrorWord._graveyard + [(ptr, dev)]
                    ErrorWord._graveyard = []
                cur = ctypes.c_int()
                have = N.lib.mifft_get_device(ctypes.byref(cur)) == 0
                for d in sorted(set(d for _, d in dead)):
                    if N.lib.mifft_set_device(d) == 0:
                        N.lib.mifft_device_sync()
                if have:
                    N.lib.mifft_set_device(cur.value)
                for p, _ in dead:
                    N.lib.mifft_host_free(p)
        except Exception:
            pass

    _graveyard = []      # (pinned word, device) released while a capture was open: freed by the next release outside one


def _foreign_capture_active():
    """A capture this module did not open itself (torch.cuda.graph(), another binding) on torch's current stream: the one place a plan
    that follows torch's stream can be recorded from."""
    try:
        import sys
        torch = sys.modules.get("torch")
        if torch is None or not torch.cuda.is_available():
            return False
        return bool(torch.cuda.is_current_stream_capturing())
    except Exception:
        return False


def _torch_current_stream(args):
    """torch's current stream if one of the buffers is a torch device tensor, else None (f1: a plan built without
    stream= runs where the producing framework runs, cuda.py:116-134 current-context semantics)."""
    for a in args:
        if type(a) is DeviceArray:
            continue
        if hasattr(a, "data_ptr") and getattr(a, "is_cuda", False):
            import torch
            return torch.cuda.current_stream(a.device)
    return None


def device_count():
    n = ctypes.c_int()
    N.check(N.lib.mifft_device_count(ctypes.byref(n)), "mifft_device_count")
    return n.value


def device_props(device=None):
    if device is None:
        d = ctypes.c_int()
        N.check(N.lib.mifft_get_device(ctypes.byref(d)), "mifft_get_device")
        device = d.value
    props = N.MifftDeviceProps()
    N.check(N.lib.mifft_device_props_get(int(device), ctypes.byref(props)), "mifft_device_props_get")
    return props


class Context(object):
    """Plan execution context (cuda.py:64-113): stream lifecycle, allocator, device limits."""

    def __init__(self, device, stream, mempool):
        # HIP has one primary context per device.  `device` None: the plan lives on the device that is current when it is
        # built.  A device index (Plan(context=i), cuda.py:121-128: the plan is built on whatever context it is given): the
        # plan's allocations, uploads and launches run with that device made current around them (activate / restore),
        # so one process can drive one plan per GPU without switching devices itself.
        if device is None:
            cur = ctypes.c_int()
            N.check(N.lib.mifft_get_device(ctypes.byref(cur)), "mifft_get_device")
            self._device = cur.value
            self._guard = False
        else:
            self._device = int(device)
            if not 0 <= self._device < device_count():
                raise ValueError("pyfft_amd: context=%d but %d device(s) are visible" % (self._device, device_count()))
            self._guard = True
        self._stream = stream
        self._call_stream = stream
        self._last_stream_handle = None     # stream of the previous enqueue (cross-stream ordering of plan scratch)
        self._order_event = None
        self._recreate_stream = stream is None
        props = device_props(self._device)
        self.device_name = props.name.decode()
        self.gcn_arch = props.gcn_arch.decode()
        self.compute_units = props.compute_units
        self.machine = Machine.from_props(props)      # what the planner sizes rings / chunks / thresholds from
        self.max_block_size = props.max_threads_per_block
        self.max_shared_mem = props.lds_bytes_per_block
        self.max_grid_x = props.max_grid_x
        self.wavefront_size = props.wavefront_size
        self._mempool = mempool
        if mempool is None:
            self.allocate = self.allocate_raw
        else:
            self.allocate = mempool.allocate

    def allocate_raw(self, nbytes):
        return DeviceAllocation(nbytes)

    def upload(self, mem, host):
        N.check(N.lib.mifft_memcpy_h2d(device_pointer(mem), host.ctypes.data, host.nbytes, None), "mifft_memcpy_h2d")

    pointer_of = staticmethod(device_pointer)

    def createQueue(self, buffers=()):
        """Stream of the coming execute(): the one given to Plan(); else torch's current stream when a buffer is a torch
        tensor; else the plan's own (blocking) stream, created on first use (cuda.py:94-96)."""
        if not self._recreate_stream:
            return
        ts = _torch_current_stream(buffers)
        if ts is not None:
            self._call_stream = ts
            return
        if self._stream is None:
            self._stream = Stream()
        self._call_stream = self._stream

    def stream_handle(self):
        return _stream_handle(self._call_stream)

    def capturing(self):
        """The stream of the coming execute() is recording into a graph (hipStreamIsCapturing)."""
        c = ctypes.c_int32()
        N.check(N.lib.mifft_stream_is_capturing(self.stream_handle(), ctypes.byref(c)), "mifft_stream_is_capturing")
        return bool(c.value)

    @property
    def device(self):
        return self._device

    def activate(self):
        """Make the plan's device current; returns what restore() needs (None when nothing was switched)."""
        if not self._guard:
            return None
        cur = ctypes.c_int()
        N.check(N.lib.mifft_get_device(ctypes.byref(cur)), "mifft_get_device")
        if cur.value == self._device:
            return None
        N.check(N.lib.mifft_set_device(self._device), "mifft_set_device")
        return cur.value

    def restore(self, prev):
        if prev is not None:
            N.check(N.lib.mifft_set_device(prev), "mifft_set_device")

    def order_scratch(self, capturing=None):
        """Called before an execute() touches plan-owned scratch (temp buffer, ring, counters): when this call runs on a
        different stream than the previous one (a plan built without stream= follows torch's current stream per call), the
        new stream first waits for the work the old one still has in flight.  The reference plan ran on ONE stream
        (cuda.py:94-107,129), where that order is implicit.  `capturing`: the caller already knows whether the stream records."""
        h = self.stream_handle()
        last = self._last_stream_handle
        if last is not None and last[0] != h:
            if capturing is None:
                capturing = self.capturing()
            if not capturing:
                if self._order_event is None:
                    self._order_event = Event()
                self._order_event.record(last[1])
                N.check(N.lib.mifft_stream_wait_event(h, self._order_event.handle), "mifft_stream_wait_event")
            # (a capturing stream cannot wait for work outside its graph, and enqueues nothing now: the caller orders the replays
            # behind the plan's earlier work, as with any captured graph -- but the capture stream becomes the plan's "last" stream, so
            # that the next eager execute on another stream waits for what has been replayed on this one by then)
        if last is None or last[0] != h:
            self._last_stream_handle = (h, self._call_stream)     # (keeps the stream object alive)

    def wait(self):
        N.check(N.lib.mifft_stream_sync(self.stream_handle()), "mifft_stream_sync")

    def flush(self):
        pass

    def getQueue(self):
        if self._call_stream is None:
            self.createQueue()
        return self._call_stream

    def isCuda(self):
        return False


def Plan(*args, **kwds):
    """Create an FFT plan (cuda.py:116-138).

    Plan(shape, dtype=numpy.complex64, mempool=None, context=None, stream=None, normalize=True,
         wait_for_finish=None, fast_math=True, scale=1.0)

    `stream`: a pyfft_amd.hip.Stream, a torch.cuda.Stream, or a raw hipStream_t value; when given,
    execute() is asynchronous by default and returns the stream.  `context`: a device index (or
    an object with a `.device` index) or None = the current device.  HIP has one primary context per device; a plan
    built for a device that is not the caller's current one makes it current around its own calls and restores the
    caller's afterwards, so one process can hold a plan per GPU (buffers must live on the plan's device).
    `mempool`: any object with an allocate(nbytes) method returning a buffer-like object.
    `parent_shape=`, `any_size=True`: opt-in extensions (tiles of a bigger array; sizes that are not powers of two), see
    pyfft_amd/generic.py.  Without them a size that is not a power of two is a ValueError, as in the reference.
    `fast_math`: accepted for signature parity and ignored -- the reference passes -use_fast_math to nvcc for its on-device
    sincos (cuda.py:56-57); here twiddles come from float64-evaluated tables, so there is nothing to relax (results meet the
    reference's thresholds either way).
    """
    mempool = kwds.pop('mempool', None)
    context_obj = kwds.pop('context', None)
    stream_obj = kwds.pop('stream', None)
    # opt-in extensions the reference lists as TODO (TODO.txt:6-8; pyfft_amd/generic.py): tiles of a parent array, any size
    parent_shape = kwds.pop('parent_shape', None)
    any_size = bool(kwds.pop('any_size', False))
    generic = parent_shape is not None or any_size
    if generic and parent_shape is None:
        # any_size=True on a power-of-two shape: the dense plan itself (no work array, no gather / scatter)
        shape = args[0] if args else kwds.get('shape')
        try:
            dims = tuple(int(v) for v in ((shape,) if isinstance(shape, (int, numpy.integer)) else shape))
            generic = not (len(dims) in (1, 2, 3) and all(v >= 1 and (v & (v - 1)) == 0 for v in dims))
        except TypeError:
            pass

    # argument errors first (ValueError, as in the reference), then the device
    if not generic:
        FFTPlan.validate(*args, **kwds)
    if device_count() < 1:
        raise RuntimeError("pyfft_amd: no HIP device visible (there is no CPU fallback)")

    # the device comes from `context` whether or not a stream is given (Plan(stream=s, context=i): the stream only decides the
    # default wait policy, cuda.py:129-134; tables, scratch and launches belong to device i -- the stream must live there too)
    device = None
    if context_obj is not None:
        if isinstance(context_obj, (int, numpy.integer)) and not isinstance(context_obj, bool):
            device = int(context_obj)
        elif hasattr(context_obj, "device"):
            device = context_obj.device
    wait_for_finish = stream_obj is None
    if stream_obj is not None and device is not None:
        sdev = getattr(stream_obj, "device_index", None)       # torch.cuda.Stream knows its device; raw handles do not
        if sdev is None:
            sdev = getattr(getattr(stream_obj, "device", None), "index", None)
        if sdev is not None and int(sdev) != int(device):
            raise ValueError("pyfft_amd: stream belongs to device %d but context=%d" % (int(sdev), int(device)))

    if 'wait_for_finish' not in kwds or kwds['wait_for_finish'] is None:
        kwds['wait_for_finish'] = wait_for_finish

    context = Context(device, stream_obj, mempool)
    prev = context.activate()       # context=i: tables and scratch are allocated on device i
    try:
        if generic:
            from .generic import GenericFFTPlan
            return GenericFFTPlan(context, *args, parent_shape=parent_shape, any_size=any_size, **kwds)
        return FFTPlan(context, *args, **kwds)
    finally:
        context.restore(prev)

"""ShardedPlan: one batch of transforms partitioned over the GPUs of a node, driven from ONE process.

The reference maps the batch onto the grid of one device (pyfft/kernel.py:99-121: `batch` blocks side by side) and binds a
plan to one context = one device (pyfft/cuda.py:67-72).  Batched transforms are independent, so the batch axis is also the
axis along which a job spreads over several devices: contiguous slices, no exchange, no collective.
Two orchestrations exist in this tree:

  * one PROCESS per GPU (bench.py under torch.distributed.run; RCCL only for the barrier and the max over ranks), and
  * this module: one process, one `Plan(context=d, stream=s_d)` per shard, every shard's execute() enqueued asynchronously
    (an execute is a few microseconds of host time and ctypes releases the GIL while the C side runs), one synchronisation per
    shard at the end.

        plan = ShardedPlan((1 << 20,), numpy.complex64, devices=[0, 1, 2, 3])
        bufs = plan.allocate(4096)                      # one DeviceArray per shard, on the shard's device
        plan.upload(bufs, host, 4096)                   # slice [start, start + count) of the host batch to every shard
        plan.execute(bufs, batch=4096)                  # in place; plan.execute(ins, outs, batch=..., inverse=True) out of place
        result = plan.download(bufs, 4096)

`devices` may name a device more than once: every entry is a shard with a plan, a stream and scratch of its own (two shards
on one device share its CUs -- what the tests do on a one-GPU box).  Buffers are the caller's, one per shard (anything Plan
accepts: DeviceArray, torch tensor on that device, raw pointer); `allocate / upload / download` are conveniences.
"""
import ctypes

import numpy


def shard_batch(global_batch, rank, world):
    """Contiguous slice [start, start + count) of the batch axis owned by shard `rank` of `world`: independent
    transforms, no exchange.  The first `global_batch % world` shards hold one transform more."""
    global_batch, rank, world = int(global_batch), int(rank), int(world)
    if world < 1 or not 0 <= rank < world or global_batch < 0:
        raise ValueError("shard_batch: need 0 <= rank < world and a non-negative batch")
    base, extra = divmod(global_batch, world)
    count = base + (1 if rank < extra else 0)
    start = rank * base + min(rank, extra)
    return start, count


class ShardError(RuntimeError):
    """One or more shards failed; `errors` = [(shard index, device, exception)] in shard order."""

    def __init__(self, errors):
        self.errors = list(errors)
        RuntimeError.__init__(self, "pyfft_amd: %d shard(s) failed: %s" % (
            len(self.errors), "; ".join("shard %d (device %d): %s" % (i, d, e) for i, d, e in self.errors)))


class _OnDevice(object):
    """`with _OnDevice(hip, d):` -- device d current inside the block, the caller's device restored afterwards."""

    def __init__(self, hip, device):
        self.hip, self.device, self.prev = hip, int(device), None

    def __enter__(self):
        N = self.hip.N
        cur = ctypes.c_int()
        N.check(N.lib.mifft_get_device(ctypes.byref(cur)), "mifft_get_device")
        if cur.value != self.device:
            N.check(N.lib.mifft_set_device(self.device), "mifft_set_device")
            self.prev = cur.value
        return self

    def __exit__(self, *exc):
        if self.prev is not None:
            self.hip.N.check(self.hip.N.lib.mifft_set_device(self.prev), "mifft_set_device")
        return False


class ShardedPlan(object):
    """`Plan()` over several devices: the batch of every execute() is cut into contiguous slices, one per shard.

    ShardedPlan(shape, dtype=numpy.complex64, devices=None, threads=False, **plan_kwds)

    devices     device indices, one per shard (default: every visible device once); repeats are allowed
    threads     enqueue every shard's execute from a host thread of its own (default: one after the other on the calling
                thread -- an asynchronous execute is launch-bound host work of a few microseconds)
    plan_kwds   what Plan() takes (normalize, scale, fast_math, mempool, any_size ...) except context / stream, which this
                class sets per shard; wait_for_finish keeps its meaning for the WHOLE sharded execute (default True, as for a
                reference plan built without a stream: pyfft/cuda.py:129-134)
    """

    def __init__(self, shape, dtype=numpy.complex64, devices=None, threads=False, _hip=None, **plan_kwds):
        if _hip is None:
            from . import hip as _hip
        self._hip = _hip
        for key in ("context", "stream"):
            if key in plan_kwds:
                raise ValueError("ShardedPlan sets %s= per shard" % key)
        self._wait_for_finish = plan_kwds.pop("wait_for_finish", None)
        if self._wait_for_finish is None:
            self._wait_for_finish = True
        ndev = _hip.device_count()
        if devices is None:
            devices = list(range(ndev))
        devices = [int(d) for d in devices]
        if not devices:
            raise ValueError("ShardedPlan needs at least one device")
        for d in devices:
            if not 0 <= d < ndev:
                raise ValueError("pyfft_amd: device %d named but %d device(s) are visible" % (d, ndev))
        self.devices = devices
        self.shape = shape
        self.dtype = numpy.dtype(dtype)
        self._split = self.dtype.kind == "f"
        self._threads = bool(threads)
        self._pool = None
        self.streams, self.plans = [], []
        for d in devices:
            with _OnDevice(_hip, d):
                s = _hip.Stream()
            self.streams.append(s)
            self.plans.append(_hip.Plan(shape, dtype=dtype, context=d, stream=s, wait_for_finish=False, **plan_kwds))

    # ---- the split of the batch axis -----------------------------------------------------------------------------------
    @property
    def nshards(self):
        return len(self.devices)

    def slices(self, global_batch):
        """[(start, count)] per shard: contiguous, in shard order, covering [0, global_batch)"""
        return [shard_batch(global_batch, i, self.nshards) for i in range(self.nshards)]

    def _items(self):
        shape = self.shape if isinstance(self.shape, tuple) else (self.shape,)
        return int(numpy.prod([int(v) for v in shape]))

    # ---- conveniences: buffers on the shards' devices --------------------------------------------------------------------
    def allocate(self, global_batch):
        """One DeviceArray of `count` transforms per shard, on the shard's device (None for an empty shard); for split-complex
        dtypes a pair (re, im) per shard."""
        out = []
        n = self._items()
        for (start, count), d in zip(self.slices(global_batch), self.devices):
            if count == 0:
                out.append(None)
                continue
            with _OnDevice(self._hip, d):
                if self._split:
                    out.append((self._hip.DeviceArray((count * n,), self.dtype), self._hip.DeviceArray((count * n,), self.dtype)))
                else:
                    out.append(self._hip.DeviceArray((count * n,), self.dtype))
        return out

    def upload(self, bufs, host, global_batch, host_im=None):
        """Slice [start, start + count) of the host batch (global_batch transforms back to back) to every shard's buffer."""
        n = self._items()
        parts = (numpy.ascontiguousarray(host).reshape(-1), None if host_im is None else numpy.ascontiguousarray(host_im).reshape(-1))
        for (start, count), d, b in zip(self.slices(global_batch), self.devices, bufs):
            if count == 0:
                continue
            with _OnDevice(self._hip, d):
                if self._split:
                    b[0].set(parts[0][start * n:(start + count) * n])
                    b[1].set(parts[1][start * n:(start + count) * n])
                else:
                    b.set(parts[0][start * n:(start + count) * n])

    def download(self, bufs, global_batch):
        """The shards' buffers gathered into one host array in batch order (split-complex: a (re, im) pair of arrays)."""
        n = self._items()
        outs = [numpy.empty(global_batch * n, self.dtype) for _ in range(2 if self._split else 1)]
        for (start, count), d, b in zip(self.slices(global_batch), self.devices, bufs):
            if count == 0:
                continue
            with _OnDevice(self._hip, d):
                if self._split:
                    outs[0][start * n:(start + count) * n] = b[0].get()
                    outs[1][start * n:(start + count) * n] = b[1].get()
                else:
                    outs[0][start * n:(start + count) * n] = b.get()
        return (outs[0], outs[1]) if self._split else outs[0]

    # ---- execution -----------------------------------------------------------------------------------------------------------
    def _fan_out(self, calls):
        """Run [(shard index, callable)] -- on the calling thread one after the other, or one host thread per shard -- and
        collect every shard's exception instead of stopping at the first (the other shards' work is already enqueued)."""
        errors = []
        if self._threads and len(calls) > 1:
            if self._pool is None:
                from concurrent.futures import ThreadPoolExecutor
                self._pool = ThreadPoolExecutor(max_workers=self.nshards)
            futures = [(i, self._pool.submit(fn)) for i, fn in calls]
            for i, f in futures:
                try:
                    f.result()
                except Exception as e:          # noqa: BLE001 - reported per shard below
                    errors.append((i, self.devices[i], e))
        else:
            for i, fn in calls:
                try:
                    fn()
                except Exception as e:          # noqa: BLE001
                    errors.append((i, self.devices[i], e))
        return errors

    def execute(self, *bufs, **kwds):
        """execute(data_in, data_out=None, inverse=False, batch=1, wait_for_finish=None)          interleaved dtypes
        execute(in_re, in_im, out_re=None, out_im=None, inverse=False, batch=1, wait_for_finish=None)   split-complex dtypes

        The reference's signatures (pyfft/plan.py:261-284) with a LIST of per-shard buffers in place of every buffer and `batch`
        = the GLOBAL batch: shard i transforms its `count_i` transforms (slices(batch)[i]) held by entry i of every list.
        Every shard's execute is enqueued asynchronously on the shard's own stream; with wait_for_finish (default: the plan's)
        the call then waits for every shard and raises ShardError if any of them failed, else it returns the list of streams."""
        inverse = kwds.pop("inverse", False)
        batch = int(kwds.pop("batch", 1))
        wait = kwds.pop("wait_for_finish", None)
        if kwds:
            raise TypeError("unexpected arguments: %s" % sorted(kwds))
        if wait is None:
            wait = self._wait_for_finish
        want = (2, 4) if self._split else (1, 2)
        if len(bufs) not in want:
            raise TypeError("execute() takes %d or %d buffer lists for dtype %s" % (want + (self.dtype.name,)))
        for b in bufs:
            if b is None or len(b) != self.nshards:
                raise ValueError("every buffer argument must be a list of %d per-shard buffers" % self.nshards)
        if batch < 1:
            raise ValueError("batch must be positive")
        calls = []
        for i, (start, count) in enumerate(self.slices(batch)):
            if count == 0:
                continue
            args = [b[i] for b in bufs]
            calls.append((i, lambda p=self.plans[i], a=args, c=count: p.execute(*a, inverse=inverse, batch=c, wait_for_finish=False)))
        errors = self._fan_out(calls)
        if wait:
            errors += [e for e in self._fan_out([(i, self.plans[i].finish) for i, _ in calls]) if e[0] not in [x[0] for x in errors]]
        if errors:
            raise ShardError(sorted(errors, key=lambda e: e[0]))
        return None if wait else list(self.streams)

    def finish(self):
        """Wait for every shard's stream; raise ShardError if any shard reported invalid results (a dependency time-out)."""
        errors = self._fan_out([(i, p.finish) for i, p in enumerate(self.plans)])
        if errors:
            raise ShardError(errors)

    def check(self):
        """Non-blocking: raise ShardError if a completed execute of any shard reported invalid results."""
        errors = self._fan_out([(i, p.check) for i, p in enumerate(self.plans)])
        if errors:
            raise ShardError(errors)

    def strategy(self, global_batch, inplace=True):
        """Every shard's execution strategy for its slice of the batch (pass-through to FFTPlan.strategy)."""
        return [p.strategy(c, inplace) if c else None for p, (_, c) in zip(self.plans, self.slices(global_batch))]

    def close(self):
        errors = self._fan_out([(i, p.close) for i, p in enumerate(self.plans)])
        if self._pool is not None:
            self._pool.shutdown(wait=True)
            self._pool = None
        if errors:
            raise ShardError(errors)

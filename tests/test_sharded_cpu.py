"""CPU tests of the library-level multi-GPU split (pyfft_amd/sharded.py): the slices of the batch axis and the fan-out of calls and
errors over the shards, against a fake `hip` module with a fake device count (no GPU, no HIP call).  The batch axis is the reference's
own unit of parallelism (pyfft/kernel.py:99-121: batch -> grid); one plan per context = per device: pyfft/cuda.py:67-72."""
import numpy
import pytest

from pyfft_amd.sharded import ShardedPlan, ShardError, shard_batch


def test_shard_batch_slices_are_contiguous_and_cover_the_batch():
    for world in (1, 2, 3, 4, 7, 8):
        for gb in (0, 1, 2, 7, 8, 9, 63, 64, 65, 4096, 65536, 65537):
            sl = [shard_batch(gb, r, world) for r in range(world)]
            assert sl[0][0] == 0 and sum(c for _, c in sl) == gb
            for (s0, c0), (s1, _) in zip(sl, sl[1:]):
                assert s1 == s0 + c0
            counts = [c for _, c in sl]
            assert max(counts) - min(counts) <= 1 and counts == sorted(counts, reverse=True)      # the first gb % world shards hold one more
    assert shard_batch(65536, 3, 8) == (3 * 8192, 8192)                                                # BASELINE.json configs[4]
    for bad in ((10, 2, 2), (10, -1, 2), (-1, 0, 1), (10, 0, 0)):
        with pytest.raises(ValueError):
            shard_batch(*bad)


class _FakeLib(object):
    def __init__(self, hip):
        self.hip = hip

    def mifft_get_device(self, ref):
        ref._obj.value = self.hip.current
        return 0

    def mifft_set_device(self, d):
        self.hip.current = int(d)
        self.hip.switches.append(int(d))
        return 0


class _FakeNative(object):
    def __init__(self, hip):
        self.lib = _FakeLib(hip)

    @staticmethod
    def check(rc, what=""):
        assert rc == 0, what


class _FakeStream(object):
    def __init__(self, hip):
        self.device = hip.current


class _FakePlan(object):
    def __init__(self, hip, shape, kw):
        self.hip, self.shape, self.kw = hip, shape, kw
        self.device = kw["context"]
        self.calls, self.finished, self.closed = [], 0, 0
        self.fail_execute = self.fail_finish = None

    def execute(self, *bufs, **kw):
        self.calls.append((bufs, kw))
        if self.fail_execute:
            raise self.fail_execute

    def finish(self):
        self.finished += 1
        if self.fail_finish:
            raise self.fail_finish

    check = finish

    def close(self):
        self.closed += 1

    def strategy(self, batch, inplace=True):
        return ("chain", batch, inplace)


class _FakeHip(object):
    def __init__(self, ndev):
        self.ndev, self.current, self.switches = ndev, 0, []
        self.N = _FakeNative(self)
        self.plans = []

    def device_count(self):
        return self.ndev

    def Stream(self):
        return _FakeStream(self)

    def Plan(self, shape, **kw):
        p = _FakePlan(self, shape, kw)
        self.plans.append(p)
        return p


def test_sharded_plan_builds_one_plan_and_stream_per_shard_on_its_device():
    hip = _FakeHip(4)
    sp = ShardedPlan((1024,), numpy.complex64, devices=[0, 1, 1, 3], _hip=hip, scale=2.0, normalize=False)
    assert sp.nshards == 4 and [p.device for p in hip.plans] == [0, 1, 1, 3]
    assert [s.device for s in sp.streams] == [0, 1, 1, 3] and hip.current == 0            # streams live on their devices, the caller's device is restored
    for p, s in zip(hip.plans, sp.streams):
        assert p.kw["stream"] is s and p.kw["wait_for_finish"] is False and p.kw["scale"] == 2.0 and p.kw["normalize"] is False
    assert ShardedPlan((16,), _hip=_FakeHip(3)).devices == [0, 1, 2]                        # default: every visible device once
    for bad in (dict(devices=[0, 4]), dict(devices=[]), dict(context=0), dict(stream=object())):
        with pytest.raises(ValueError):
            ShardedPlan((16,), _hip=_FakeHip(4), **bad)


def test_sharded_execute_cuts_the_batch_and_fans_errors_out():
    hip = _FakeHip(2)
    sp = ShardedPlan((64,), numpy.complex64, devices=[0, 1, 0], _hip=hip)
    ins, outs = ["i0", "i1", "i2"], ["o0", "o1", "o2"]
    assert sp.execute(ins, outs, batch=10, inverse=True) is None                           # waits by default (the reference's rule without a stream)
    assert [p.calls for p in hip.plans] == [[(("i0", "o0"), dict(inverse=True, batch=4, wait_for_finish=False))],
                                            [(("i1", "o1"), dict(inverse=True, batch=3, wait_for_finish=False))],
                                            [(("i2", "o2"), dict(inverse=True, batch=3, wait_for_finish=False))]]
    assert [p.finished for p in hip.plans] == [1, 1, 1]
    assert sp.execute(ins, batch=2, wait_for_finish=False) == sp.streams                    # asynchronous: the streams; the empty shard is skipped
    assert [len(p.calls) for p in hip.plans] == [2, 2, 1] and [p.finished for p in hip.plans] == [1, 1, 1]
    assert sp.strategy(2) == [("chain", 1, True), ("chain", 1, True), None]
    # a shard that fails to enqueue: the others are enqueued and waited for all the same, the error names the shard
    hip.plans[1].fail_execute = RuntimeError("boom")
    with pytest.raises(ShardError) as ei:
        sp.execute(ins, outs, batch=9)
    assert [(i, d) for i, d, _ in ei.value.errors] == [(1, 1)] and "shard 1 (device 1): boom" in str(ei.value)
    assert [len(p.calls) for p in hip.plans] == [3, 3, 2] and [p.finished for p in hip.plans] == [2, 2, 2]
    hip.plans[1].fail_execute = None
    # a dependency time-out reported by finish() of two shards: both are listed, in shard order
    hip.plans[0].fail_finish = RuntimeError("time-out")
    hip.plans[2].fail_finish = RuntimeError("time-out")
    with pytest.raises(ShardError) as ei:
        sp.finish()
    assert [i for i, _, _ in ei.value.errors] == [0, 2]
    with pytest.raises(ShardError):
        sp.check()
    with pytest.raises(ShardError):
        sp.execute(ins, batch=9)
    hip.plans[0].fail_finish = hip.plans[2].fail_finish = None
    # argument shape: one list per buffer of the reference's signature, one entry per shard
    for bad in ((ins[:2],), (ins, outs, ins), ()):
        with pytest.raises((ValueError, TypeError)):
            sp.execute(*bad, batch=4)
    with pytest.raises(ValueError):
        sp.execute(ins, batch=0)
    with pytest.raises(TypeError):
        sp.execute(ins, batch=4, bogus=1)
    sp.close()
    assert [p.closed for p in hip.plans] == [1, 1, 1]


def test_sharded_split_layout_and_threads():
    hip = _FakeHip(2)
    sp = ShardedPlan((8, 8), numpy.float32, devices=[0, 1], threads=True, _hip=hip)
    re, im = ["r0", "r1"], ["m0", "m1"]
    sp.execute(re, im, batch=5)                                                            # in place, split planes: two lists
    assert hip.plans[0].calls == [(("r0", "m0"), dict(inverse=False, batch=3, wait_for_finish=False))]
    assert hip.plans[1].calls == [(("r1", "m1"), dict(inverse=False, batch=2, wait_for_finish=False))]
    with pytest.raises(TypeError):
        sp.execute(re, batch=5)                                                            # a split plan takes 2 or 4 lists
    sp.execute(re, im, re, im, batch=5)
    assert len(hip.plans[0].calls) == 2 and len(hip.plans[0].calls[1][0]) == 4
    sp.close()

"""GPU tests of what surrounds the kernels (b, e, f1, f2): the reference's quick-start literally, stream ordering
against torch producers, Plan(context=i / stream=s), device properties, executes recorded into hipGraphs (hip.Graph, torch.cuda.graph()), four
host threads side by side, the batch-sharded path with eight ranks and -- round 6 -- from one process (pyfft_amd.sharded.ShardedPlan), the vendor
comparator as a value cross-check and the published-table benchmark."""
import ctypes
import json
import os
import subprocess
import sys

import numpy
import pytest

import pyfft_oracle as oracle
from helpers import EPS_F, MAX_F, _noise, _test_data, _tiled_noise

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_quick_start_literally(ctx):
    """doc/source/index.rst:61-92: `Plan(..., stream=s); plan.execute(g); g.get()` with NO synchronize in between.  The
    plan's stream is a blocking stream, so the null-stream copy of get() orders behind the transform, as with PyCUDA."""
    n = 1 << 22                                          # long enough that an unordered copy would read stale data
    data = numpy.ones(n, dtype=numpy.complex64)
    stream = ctx.hip.Stream()
    plan = ctx.getPlan((n,), dtype=numpy.complex64, stream=stream)
    for _ in range(3):
        gpu_data = ctx.toGpu(data)
        plan.execute(gpu_data)                           # asynchronous: a stream was given
        result = gpu_data.get()                          # no stream.synchronize()
        assert abs(result[0] - n) < 1e-3 * n and numpy.abs(result[1:]).max() < 1e-3 * n
        plan.execute(gpu_data, inverse=True)
        assert numpy.abs(gpu_data.get() - data).max() < 1e-4


def test_default_plan_orders_against_torch_producer(ctx):
    """f1 (cuda.py:116-134 current-context semantics): a plan built without stream= runs on torch's current stream when the
    buffers are torch tensors, so a torch kernel that produces the input and execute() are ordered -- on torch's default
    stream and on a side stream alike -- without any explicit synchronisation."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("torch sees no GPU")
    n, batch = 1 << 20, 64
    plan = ctx.getPlan((n,), dtype=numpy.complex64)
    for use_side in (False, True):
        s = torch.cuda.Stream() if use_side else torch.cuda.current_stream()
        with torch.cuda.stream(s):
            for rep in range(3):
                x = torch.zeros(batch, n, dtype=torch.complex64, device="cuda")
                y = torch.empty_like(x)
                x += (rep + 1.0)                          # producer kernel on the current stream, enqueued just before
                ret = plan.execute(x, y, batch=batch, wait_for_finish=False)
                assert ret is not None and int(ret.cuda_stream) == int(s.cuda_stream)
                z = y[:, :2].clone()                      # consumer kernel on the same stream
                s.synchronize()
                assert torch.allclose(z[:, 0].real, torch.full((batch,), (rep + 1.0) * n, device="cuda"), rtol=1e-5)
                assert z[:, 1].abs().max().item() < 1e-2
    plan.finish()


def test_context_device_must_be_current(ctx):
    ctx.getPlan((16,), dtype=numpy.complex64, context=0)
    with pytest.raises(ValueError):
        ctx.getPlan((16,), dtype=numpy.complex64, context=ctx.hip.device_count() + 3)


def test_bench_nccl_path_at_world_size_one():
    """bench.py --force-dist: the RCCL process group, barrier and max-over-ranks reduction really run (world size 1),
    on a reduced batch; the line keeps its contract fields."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--batch", "128", "--steps", "3",
                          "--warmup", "1", "--no-cpu-baseline", "--repeats", "2"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1
    res = json.loads(line[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "protocol"):
        assert key in res
    assert res["n_gpus"] == 1 and res["parity"]["ok"] and res["roofline"]["bound"] == "hbm"
    assert res["protocol"]["in_place"]["ms_per_execute"]["n"] == 2


def test_plan_close_releases_and_plan_stays_usable(ctx):
    n, batch = 1 << 16, 64
    plan = ctx.getPlan((n,), dtype=numpy.complex64, wait_for_finish=False)
    a = ctx.toGpu(numpy.ones(n * batch, dtype=numpy.complex64))
    plan.execute(a, batch=batch)
    plan.close()
    assert plan._tempmemobj is None and plan._side_streams is None
    plan.execute(a, inverse=True, batch=batch)
    plan.finish()
    assert numpy.abs(a.get() - 1).max() < 1e-5


# ---- the sharded path with more than one rank --------------------------------------------------------------------------
@pytest.mark.skipif(not os.environ.get("PYFFT_AMD_SWEEP"), reason="superseded by tests/test_interop_gpu.py::test_eight_ranks_share_one_gpu (the same path with eight ranks); runs with the soak switch")
def test_two_ranks_share_one_gpu_sharded_path(tmp_path):
    """`bench.py --gpus 2` for real: two processes (one plan, stream and scratch each) on ONE device, gloo as the control
    plane, data taken from the GLOBAL dataset by transform index.  Each rank parity-checks its slice [start, start + count)
    in-process; here the first and last transform of every slice are checked again, against numpy on the global dataset
    regenerated independently of any rank."""
    sys.path.insert(0, ROOT)
    import bench
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    batch = 96                                                # per rank: 2 x 96 transforms of 8 MiB
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--control", "gloo", "--share-gpu",
                          "--config", "c2", "--batch", str(batch), "--steps", "2", "--warmup", "1", "--plain",
                          "--dump-dir", str(tmp_path)],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["config"]["global_batch"] == 2 * batch and res["scaling"] == "weak"
    ranks = res["config"]["ranks"]
    assert [r["first_transform"] for r in ranks] == [0, batch] and [r["count"] for r in ranks] == [batch, batch]
    assert all(r["parity_ok"] for r in ranks) and all(r["device"] == 0 for r in ranks)
    assert res["config"]["strategy"] == "fused2"              # the persistent kernel of the headline path, in both processes
    shape, dtname, _, seed = bench.CONFIGS["c2"]
    for g in (0, batch - 1, batch, 2 * batch - 1):
        got = numpy.load(os.path.join(str(tmp_path), "xform_%d.npy" % g))
        ref = numpy.fft.fft(bench.global_item(shape, dtname, batch, seed, g).astype(numpy.complex128))
        assert numpy.abs(got - ref).sum() / numpy.abs(ref).sum() < 1.1e-6, g
        assert numpy.abs(got - ref).max() <= 1e-5 * numpy.abs(ref).max(), g
    # weak scaling arithmetic of the line: value = all ranks' transforms over the max-over-ranks time
    assert abs(res["transforms_per_s"] - 2 * batch * res["steps"] / (res["ms_per_step"] * 1e-3 * res["steps"])) < 1e-6 * res["transforms_per_s"]


# ---- f2: the vendor comparator as a value cross-check, and the published-table benchmark ------------------------------
def _run_tool(args, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    return subprocess.run([sys.executable] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def test_hipfft_and_libmifft_agree_on_the_comparator_shapes():
    """tools/hipfft_check.py (cuda/test.cu:13-95 counterpart): hipFFT and libmifft transform the same seeded device buffer;
    values agree within the reference's thresholds on the eight comparator shapes, and both agree with numpy."""
    out = _run_tool([os.path.join(ROOT, "tools", "hipfft_check.py")])
    if out.returncode == 2:
        pytest.skip("no hipFFT library on this box")
    assert out.returncode == 0, (out.stdout[-3000:], out.stderr[-2000:])
    lines = [l for l in out.stdout.splitlines() if "L1-rel" in l]
    assert len(lines) == 8 and all(l.rstrip().endswith("ok") for l in lines), out.stdout


def test_perf_table_quick_uses_the_reference_formula():
    """tools/perf_table.py --quick (test/test_performance.py:11,22-30): batch fills the 32 MiB buffer, GFLOPS =
    5e-9 * sum(log2 dims) * points * batch / t, and the numbers are in a sane range for this part."""
    out = _run_tool([os.path.join(ROOT, "tools", "perf_table.py"), "--quick"])
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-2000:])
    rows = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert [tuple(r["shape"]) for r in rows] == [(1024,), (128, 128)]
    for r in rows:
        size = int(numpy.prod(r["shape"]))
        assert r["batch"] == (32 << 20) // (size * 8)
        want = 5.0e-9 * sum(numpy.log2(s) for s in r["shape"]) * size * r["batch"] / r["seconds_per_execute"]
        assert abs(r["gflops"] - want) < 1e-6 * want
        # one HBM round trip of a 32 MiB buffer: between 2 % and 100 % of the 8 TB/s roofline
        frac = 2.0 * size * 8 * r["batch"] / r["seconds_per_execute"] / 8e12
        assert 0.02 < frac < 1.0, frac


def test_plan_following_two_torch_streams_without_host_sync(ctx):
    """A plan built without stream= runs each execute() on torch's CURRENT stream.  Alternating two torch streams with no
    host synchronisation in between must not let the second launch's counter reset / ring writes race with the first
    persistent kernel: the plan orders the new stream behind the old one on its scratch (Context.order_scratch)."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("torch sees no GPU")
    n, batch = 1 << 18, 160                                   # fused2 (persistent, ring + counters owned by the plan; > 256 MiB per side)
    plan = ctx.getPlan((n,), dtype=numpy.complex64, wait_for_finish=False)
    assert plan.strategy(batch)[0] == "fused2"
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    g = torch.Generator(device="cpu").manual_seed(3)
    host = torch.randn(2, n, dtype=torch.complex64, generator=g)
    xs = [host[i].repeat(batch, 1).cuda() for i in range(2)]
    ys = [torch.empty_like(x) for x in xs]
    torch.cuda.synchronize()
    for rep in range(6):
        for i, s in enumerate((s1, s2)):
            with torch.cuda.stream(s):
                ret = plan.execute(xs[i], ys[i], batch=batch)
                assert int(ret.cuda_stream) == int(s.cuda_stream)
    torch.cuda.synchronize()
    plan.finish()
    for i in range(2):
        ref = numpy.fft.fft(host[i].numpy().astype(numpy.complex128))
        got = ys[i].cpu().numpy()
        for b in (0, 1, batch // 2, batch - 1):
            assert numpy.abs(got[b] - ref).sum() / numpy.abs(ref).sum() < 1.1e-6, (i, b)


def test_plan_for_a_device_given_by_index(ctx):
    """Plan(context=i) (cuda.py:121-128: the plan is built on whatever context it is given): the plan makes device i current
    around its own calls and restores the caller's.  With one visible device this exercises the guard with i == current; with
    more, a plan per device is driven from one process without the caller switching devices."""
    hip = ctx.hip
    N = hip.N
    import ctypes
    ndev = hip.device_count()
    with pytest.raises(ValueError):
        ctx.getPlan((1024,), dtype=numpy.complex64, context=ndev)      # not a visible device
    data = oracle.get_test_data((4096,), numpy.complex64, 4, 9)
    ref = oracle.numpy_fft(numpy.fft.fftn, data, 4)
    cur = ctypes.c_int()
    N.check(N.lib.mifft_get_device(ctypes.byref(cur)), "get")
    home = cur.value
    for dev in range(ndev):
        plan = ctx.getPlan((4096,), dtype=numpy.complex64, context=dev)
        assert plan._context.device == dev and plan._context._guard
        N.check(N.lib.mifft_get_device(ctypes.byref(cur)), "get")
        assert cur.value == home                               # construction restored the caller's device
        N.check(N.lib.mifft_set_device(dev), "set")            # buffers live on the plan's device
        a = ctx.toGpu(data)
        N.check(N.lib.mifft_set_device(home), "set")
        plan.execute(a, batch=4)                               # called with `home` current
        N.check(N.lib.mifft_get_device(ctypes.byref(cur)), "get")
        assert cur.value == home
        N.check(N.lib.mifft_set_device(dev), "set")
        got = a.get()
        N.check(N.lib.mifft_set_device(home), "set")
        assert oracle.difference(ref, got, 4) < 1.1e-6


def test_device_properties_describe_the_memory_system(ctx):
    """mifft_device_props carries what the planner needs (include/mifft.h): on an MI355X 256 CUs in 8 XCDs with 4 MiB of L2
    each and the 256 MiB Infinity Cache -- read from the HSA agent, not hard-wired."""
    props = ctx.hip.device_props()
    m = ctx.hip.Machine.from_props(props)
    assert props.compute_units >= 1 and props.num_xcc >= 1 and props.llc_bytes >= 0
    if props.gcn_arch.decode().startswith("gfx950") and props.compute_units == 256:
        assert props.num_xcc == 8 and props.llc_bytes == 256 << 20 and props.l2_bytes == 4 << 20, (props.num_xcc, props.llc_bytes, props.l2_bytes)
        assert m.xcd_cooperative and m.ring_bytes == 224 << 20


def test_plan_with_stream_and_context_index(ctx):
    """Plan(stream=s, context=i): the device comes from `context` also when a stream is given (cuda.py:121-134); the plan is
    asynchronous by default and guarded for device i.  With several GPUs the last one is used from device 0."""
    hip = ctx.hip
    N = hip.N
    ndev = hip.device_count()
    dev = ndev - 1
    cur = ctypes.c_int()
    N.check(N.lib.mifft_get_device(ctypes.byref(cur)), "get")
    home = cur.value
    N.check(N.lib.mifft_set_device(dev), "set")
    stream = hip.Stream()
    data = _test_data((8192,), numpy.complex64, 3, 95)
    a = ctx.toGpu(data)
    N.check(N.lib.mifft_set_device(home), "set")
    plan = ctx.getPlan((8192,), dtype=numpy.complex64, stream=stream, context=dev)
    assert plan._context.device == dev and plan._context._guard and plan._wait_for_finish is False
    assert plan.execute(a, batch=3) is stream
    plan.finish()
    N.check(N.lib.mifft_get_device(ctypes.byref(cur)), "get")
    assert cur.value == home
    N.check(N.lib.mifft_set_device(dev), "set")
    got = a.get()
    N.check(N.lib.mifft_set_device(home), "set")
    assert oracle.difference(oracle.numpy_fft(numpy.fft.fft, data, 3), got, 3) < 1.1e-6

    class FakeTorchStream(object):                 # a stream that knows its device (torch.cuda.Stream.device_index)
        cuda_stream = stream.handle
        device_index = dev + 1
    with pytest.raises(ValueError, match="stream belongs to device"):
        ctx.getPlan((8192,), dtype=numpy.complex64, stream=FakeTorchStream(), context=dev)


def test_eight_ranks_share_one_gpu(tmp_path):
    """`bench.py --gpus 8` for real (the harness had never started more than two ranks): eight processes, each with its own plan,
    stream and scratch, on ONE device (--share-gpu), gloo as the control plane, each rank's slice of the global batch checked
    in-process and its first / last transform again here against numpy on the global data set.  rank 0 also times the CPU baseline
    (numpy.fft on the host cores in the same run, at every world size)."""
    sys.path.insert(0, ROOT)
    import bench
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    batch = 40                                                # per rank: 320 MiB per side -- the persistent kernel of the headline path in every process
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--control", "gloo", "--share-gpu",
                          "--config", "c2", "--batch", str(batch), "--steps", "2", "--warmup", "1", "--repeats", "0",
                          "--cpu-budget", "1", "--cpu-workers", "16", "--dump-dir", str(tmp_path)],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 8 and res["config"]["global_batch"] == 8 * batch and res["scaling"] == "weak"
    ranks = res["config"]["ranks"]
    assert [r["rank"] for r in ranks] == list(range(8))
    assert [r["first_transform"] for r in ranks] == [batch * r for r in range(8)] and all(r["count"] == batch for r in ranks)
    assert all(r["parity_ok"] for r in ranks) and all(r["device"] == 0 for r in ranks)
    assert res["config"]["strategy"] == "fused2"
    cpu = res["cpu_baseline"]
    assert cpu is not None and cpu["kind"] == "reference" and cpu["value"] > 0 and cpu["best"]["value"] >= cpu["value"]
    shape, dtname, _, seed = bench.CONFIGS["c2"]
    for r in range(8):
        for g in (r * batch, (r + 1) * batch - 1):
            got = numpy.load(os.path.join(str(tmp_path), "xform_%d.npy" % g))
            ref = numpy.fft.fft(bench.global_item(shape, dtname, batch, seed, g).astype(numpy.complex128))
            assert numpy.abs(got - ref).sum() / numpy.abs(ref).sum() < EPS_F, g
            assert numpy.abs(got - ref).max() <= MAX_F * numpy.abs(ref).max(), g
    assert abs(res["transforms_per_s"] - 8 * batch * res["steps"] / (res["ms_per_step"] * 1e-3 * res["steps"])) < 1e-6 * res["transforms_per_s"]


CAPTURE_CASES = [((1 << 18,), 160, numpy.complex64, "fused2"),        # 28 / 56 ring: the review's case
                 ((1 << 20,), 64, numpy.complex64, "fused2"),         # BASELINE config 2's kernel
                 ((128, 128, 128), 32, numpy.complex64, "fusedp"),
                 ((1024, 1024), 64, numpy.complex64, "fused2"),       # the 2-D form
                 ((1 << 16,), 96, numpy.complex64, "chain"),
                 ((512, 1024), 80, numpy.complex128, "fused2"),
                 ((256, 4096), 40, numpy.complex64, "pipelined"),     # cache-sized chunks: eager on the plan's side streams, recorded as a linear graph
                 ((128, 512, 512), 2, numpy.complex64, "fused2z")]    # round 6: persistent launch over the planes + a plain z launch in one graph


@pytest.mark.parametrize("shape,batch,dtype,strategy", CAPTURE_CASES, ids=lambda v: getattr(v, "__name__", str(v)))
def test_captured_execute_replays_bit_identically(ctx, shape, batch, dtype, strategy):
    """One execute() recorded into a hipGraph (hip.Graph: mifft_stream_begin_capture / _end_capture) and replayed five times,
    with eager executes of the same plan in between: every replay writes the bits of the eager result over the WHOLE array and the
    error word stays clean.  (pyfft/plan.py:250-259: execute is an asynchronous enqueue on the caller's stream -- which a caller
    may capture.  The two alternating counter sets of the persistent launches are host state: a captured launch has its own set
    with the memset as a graph node.)"""
    from pyfft_amd import _native as N
    hip = ctx.hip
    size = int(numpy.prod(shape))
    data = _tiled_noise(size * batch, dtype, 501)
    s = hip.Stream()
    plan = hip.Plan(shape if len(shape) > 1 else shape[0], dtype=dtype, stream=s)
    a = hip.to_gpu(data)
    b = hip.DeviceArray((size * batch,), dtype)
    assert plan.strategy(batch)[0] == strategy, plan.strategy(batch)
    plan.execute(a, b, batch=batch)
    s.synchronize()
    want = b.get().view(numpy.uint32)
    N.check(N.lib.mifft_memset(b.ptr, 0, b.nbytes, s.handle))
    with hip.Graph(s) as g:
        assert plan._context.capturing()
        assert plan.execute(a, b, batch=batch) is s
    assert not plan._context.capturing()
    s.synchronize()
    assert not b.get().any(), "a captured execute ran"
    for i in range(5):
        N.check(N.lib.mifft_memset(b.ptr, 0, b.nbytes, s.handle))
        g.launch()
        if i in (1, 2):
            g.launch()                                  # back to back on the same counter set
        s.synchronize()
        assert numpy.array_equal(b.get().view(numpy.uint32), want), ("replay", i)
        if i in (0, 3):                                 # eager executes in between keep alternating their own two sets
            N.check(N.lib.mifft_memset(b.ptr, 0, b.nbytes, s.handle))
            plan.execute(a, b, batch=batch)
            if i == 3:
                plan.execute(a, b, batch=batch)
            s.synchronize()
            assert numpy.array_equal(b.get().view(numpy.uint32), want), ("eager after replay", i)
    plan.finish()                                       # raises if any launch reported a dependency time-out
    # a graph must survive whatever else the process does with the library -- building ANOTHER plan (tables allocated and uploaded)
    # broke every later replay of a persistent launch under the HIP runtime PyTorch bundles while the counter set was zeroed by a
    # memset NODE (round 6; the set is zeroed by a kernel of the library's own now, csrc/fft_aux.hip)
    bystander = hip.Plan(64, dtype=numpy.complex64, stream=s)
    N.check(N.lib.mifft_memset(b.ptr, 0, b.nbytes, s.handle))
    g.launch()
    s.synchronize()
    assert numpy.array_equal(b.get().view(numpy.uint32), want), "replay after another plan was built"
    del bystander

    # a batch the plan has not run yet cannot be captured when its scratch would be allocated inside the capture: loud, not wrong
    # (round 6: a plan whose chain runs in place throughout -- (256, 4096): ROW + in-place COL -- owns no scratch, and may)
    if plan._temp_buffer_needed or strategy in ("fused2", "fusedp"):
        with pytest.raises(RuntimeError, match="eager execute"):
            with hip.Graph(s):
                plan.execute(a, b, batch=batch - 1)
        assert not plan._context.capturing()            # (the failed body left no capture behind)
    else:
        # (what the smaller batch computes eagerly: the same bits as `want` where it runs the same kernels -- the plane-fused route of
        # the last case gives way to the plain chain at batch 1, other kernels for x and y)
        if strategy == "fused2z":
            import os
            os.environ["PYFFT_AMD_NO_PLANE_FUSED"] = "1"
            try:
                other = hip.Plan(shape, dtype=dtype, stream=s)
                other.execute(a, b, batch=batch)
                s.synchronize()
            finally:
                del os.environ["PYFFT_AMD_NO_PLANE_FUSED"]
            want1 = b.get().view(numpy.uint32).reshape(batch, -1)
        else:
            want1 = want.reshape(batch, -1)
        N.check(N.lib.mifft_memset(b.ptr, 0, b.nbytes, s.handle))
        with hip.Graph(s) as g2:
            plan.execute(a, b, batch=batch - 1)
        g2.launch()
        s.synchronize()
        got = b.get().view(numpy.uint32).reshape(batch, -1)
        assert numpy.array_equal(got[:batch - 1], want1[:batch - 1]) and not got[batch - 1].any()
    # executes that wait cannot be recorded (waiting synchronises the stream): a clear error, and the capture ends cleanly
    with pytest.raises(RuntimeError, match="cannot wait"):
        with hip.Graph(s):
            plan.execute(a, b, batch=batch, wait_for_finish=True)
    assert not plan._context.capturing()
    # LIFETIME (hip.Graph docstring): the graph keeps the plan alive, the plan keeps the scratch the graph replays on -- another batch
    # re-prepares the plan, close() releases what it can, the last reference to the plan goes away; the first graph still replays
    assert any(p is plan for p in g._plans)
    plan.execute(a, b, batch=max(1, batch // 2))
    plan.close()
    del plan
    import gc
    gc.collect()
    N.check(N.lib.mifft_memset(b.ptr, 0, b.nbytes, s.handle))
    g.launch()
    s.synchronize()
    assert numpy.array_equal(b.get().view(numpy.uint32), want), "replay after the plan was re-prepared, closed and dropped"
    s.synchronize()


THREAD_CASES = [((1 << 18,), 160, numpy.complex64, "fused2"), ((128, 128, 128), 32, numpy.complex64, "fusedp"),
                ((1024, 1024), 40, numpy.complex128, "fused2"), ((16, 16, 128), 512, numpy.complex64, "chain")]


def test_plans_of_four_host_threads_run_side_by_side(ctx):
    """The library keeps no mutable state between plans (re-entrant): four host threads, each with a plan, a stream,
    scratch and counters of its own -- three of them persistent launches, which then share the CUs and poll their own counters -- execute
    concurrently, six executes each, alternating out of place and in place.  Every thread's results carry the bits of the same plan run
    alone, the error words stay clean.  (One PLAN is one thread's at a time, as the reference's: pyfft/plan.py:200-259 keeps the temp
    buffer in the plan.)"""
    import threading
    hip = ctx.hip
    jobs = []
    for i, (shape, batch, dtype, strategy) in enumerate(THREAD_CASES):
        size = int(numpy.prod(shape))
        data = _tiled_noise(size * batch, dtype, 700 + i)
        st = hip.Stream()
        plan = hip.Plan(shape if len(shape) > 1 else shape[0], dtype=dtype, stream=st)
        assert plan.strategy(batch)[0] == strategy, plan.strategy(batch)
        a, b = hip.to_gpu(data), hip.DeviceArray((size * batch,), dtype)
        plan.execute(a, b, batch=batch)
        st.synchronize()
        want = b.get().view(numpy.uint32)
        plan.execute(b, batch=batch, inverse=True)          # (in place, the other direction: the second thing every thread does)
        st.synchronize()
        back = b.get().view(numpy.uint32)
        jobs.append(dict(plan=plan, stream=st, a=a, b=b, batch=batch, want=want, back=back, errors=[]))
    gate = threading.Barrier(len(jobs))

    def work(j):
        try:
            gate.wait(timeout=60)
            for rep in range(3):
                j["plan"].execute(j["a"], j["b"], batch=j["batch"])
                j["stream"].synchronize()
                if not numpy.array_equal(j["b"].get().view(numpy.uint32), j["want"]):
                    j["errors"].append(("forward", rep))
                j["plan"].execute(j["b"], batch=j["batch"], inverse=True)
                j["stream"].synchronize()
                if not numpy.array_equal(j["b"].get().view(numpy.uint32), j["back"]):
                    j["errors"].append(("inverse in place", rep))
            j["plan"].finish()
        except Exception as e:                               # (a worker's exception must fail the test, not vanish with the thread)
            j["errors"].append(repr(e))

    threads = [threading.Thread(target=work, args=(j,)) for j in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in threads), "a worker hangs"
    assert [j["errors"] for j in jobs] == [[] for _ in jobs]


def test_torch_cuda_graph_around_execute(ctx):
    """torch.cuda.graph() around execute() of a plan built without stream= (it follows torch's current stream, cuda.py:116-134):
    the persistent kernel of 2^18 x 160, replayed on new input values."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("torch sees no GPU")
    hip = ctx.hip
    n, batch = 1 << 18, 160
    x = torch.view_as_complex(torch.randn(batch * n, 2, device="cuda", dtype=torch.float32))
    y = torch.empty_like(x)
    plan = hip.Plan(n, dtype=numpy.complex64, wait_for_finish=False)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        plan.execute(x, y, batch=batch)                 # warm-up on the side stream, as torch's graph recipe prescribes
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    assert plan.strategy(batch)[0] == "fused2"
    want = torch.fft.fft(x.view(batch, n), dim=1).reshape(-1)
    assert (y - want).abs().sum() / want.abs().sum() < 2e-6
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        plan.execute(x, y, batch=batch)
    for k in range(3):
        x.copy_(torch.view_as_complex(torch.randn(batch * n, 2, device="cuda", dtype=torch.float32)))
        y.zero_()
        g.replay()
        torch.cuda.synchronize()
        want = torch.fft.fft(x.view(batch, n), dim=1).reshape(-1)
        assert (y - want).abs().sum() / want.abs().sum() < 2e-6, k
    plan.finish()


# ---- the batch split as library code, driven from ONE process (pyfft_amd/sharded.py; round 6) --------------------------------------------
SHARDED_CASES = [((1 << 20,), numpy.complex64, 96), ((128, 128, 128), numpy.complex64, 70), ((1024, 1024), numpy.float32, 75), ((4096,), numpy.complex128, 1001)]


@pytest.mark.parametrize("shape,dtype,batch", SHARDED_CASES, ids=lambda v: getattr(v, "__name__", str(v)).replace(" ", ""))
def test_sharded_plan_two_shards_on_one_device(ctx, shape, dtype, batch):
    """ShardedPlan(devices=[0, 0]): two shards -- a plan, a stream and scratch each -- on ONE device, the global batch cut into two
    contiguous slices (an odd batch: 38 + 37 ...), both executes enqueued asynchronously from one process and waited for once.  The
    whole result against ONE ordinary plan on the global batch (the same kernels on other slices: the reference's thresholds; bit
    identity where both run one strategy), first / last transform of EVERY shard against numpy on the global data set, out of place
    with the input untouched, in place, the inverse.  (pyfft/kernel.py:99-121: batch -> grid; pyfft/cuda.py:67-72: one device per plan.)"""
    from pyfft_amd.sharded import ShardedPlan
    hip = ctx.hip
    cdt = numpy.dtype(dtype)
    split = cdt.kind == "f"
    double = cdt in (numpy.dtype(numpy.complex128), numpy.dtype(numpy.float64))
    eps, mx = (1e-11, 1e-10) if double else (EPS_F, MAX_F)
    size = int(numpy.prod(shape))
    fdt = numpy.float64 if double else numpy.float32
    rng = numpy.random.default_rng(6100 + size % 97)
    re = _noise(rng, size * batch, fdt)
    im = _noise(rng, size * batch, fdt)
    x = re.astype(numpy.complex128) + 1j * im
    sp = ShardedPlan(shape if len(shape) > 1 else shape[0], dtype, devices=[0, 0])
    assert sp.nshards == 2 and sp.slices(batch) == [(0, batch - batch // 2), (batch - batch // 2, batch // 2)]
    assert sp.streams[0].handle != sp.streams[1].handle and sp.plans[0] is not sp.plans[1]
    host = (re, im) if split else ((re + 1j * im).astype(cdt),)
    ins, outs = sp.allocate(batch), sp.allocate(batch)
    if split:
        sp.upload(ins, re, batch, host_im=im)
    else:
        sp.upload(ins, host[0], batch)
    args_in = ([b[0] for b in ins], [b[1] for b in ins]) if split else (ins,)
    args_out = ([b[0] for b in outs], [b[1] for b in outs]) if split else (outs,)
    assert sp.execute(*(args_in + args_out), batch=batch) is None            # waits: the reference's rule without a stream
    got = sp.download(outs, batch)
    got = (got[0].astype(numpy.complex128) + 1j * got[1]) if split else got.astype(numpy.complex128)
    back_in = sp.download(ins, batch)
    assert all(numpy.array_equal(a, b) for a, b in zip(back_in if split else (back_in,), host)), "an out-of-place execute touched its input"
    for start, count in sp.slices(batch):
        for g in (start, start + count - 1):
            ref = numpy.fft.fftn(x[g * size:(g + 1) * size].reshape(shape)).reshape(-1)
            d = got[g * size:(g + 1) * size]
            assert numpy.abs(ref - d).sum() / numpy.abs(ref).sum() < eps and numpy.abs(ref - d).max() <= mx * numpy.abs(ref).max(), g
    # one ordinary plan on the global batch
    plan = hip.Plan(shape if len(shape) > 1 else shape[0], dtype=dtype)
    a = [hip.to_gpu(h) for h in host]
    b = [hip.DeviceArray(h.shape, h.dtype) for h in host]
    plan.execute(*(a + b), batch=batch)
    one = [t.get() for t in b]
    one = (one[0].astype(numpy.complex128) + 1j * one[1]) if split else one[0].astype(numpy.complex128)
    assert numpy.abs(one - got).sum() / numpy.abs(one).sum() < (1e-14 if double else 5e-7)
    # in place, asynchronously, then the inverse in place: the round trip
    streams = sp.execute(*args_in, batch=batch, wait_for_finish=False)
    assert streams == sp.streams
    sp.finish()
    inp = sp.download(ins, batch)
    inp = (inp[0].astype(numpy.complex128) + 1j * inp[1]) if split else inp.astype(numpy.complex128)
    assert numpy.array_equal(inp, got), "in place differs from out of place"
    sp.execute(*args_in, batch=batch, inverse=True)
    rt = sp.download(ins, batch)
    rt = (rt[0].astype(numpy.complex128) + 1j * rt[1]) if split else rt.astype(numpy.complex128)
    assert numpy.abs(rt - x).sum() / numpy.abs(x).sum() < eps
    assert [s_[0] for s_ in sp.strategy(batch)] == [plan.strategy(c)[0] for _, c in sp.slices(batch)]
    sp.check()
    sp.close()


def test_bench_single_process_four_shards_share_one_gpu(tmp_path):
    """`bench.py --gpus 4 --single-process --share-gpu`: the sharded job driven from ONE process through ShardedPlan (four shards on one
    device here), the same JSON line as one process per GPU; every shard parity-checked in-process, its first / last transform again here
    against numpy on the global data set regenerated independently."""
    sys.path.insert(0, ROOT)
    import bench
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    batch = 40
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--single-process", "--share-gpu", "--config", "c2",
                          "--batch", str(batch), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--dump-dir", str(tmp_path)],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    res = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in res
    assert res["n_gpus"] == 4 and res["steps"] == 3 and res["config"]["global_batch"] == 4 * batch and res["scaling"] == "weak"
    assert "ONE process" in res["config"]["parallelism"] and res["config"]["devices"] == [0, 0, 0, 0]
    ranks = res["config"]["ranks"]
    assert [r["first_transform"] for r in ranks] == [batch * r for r in range(4)] and all(r["count"] == batch and r["parity_ok"] for r in ranks)
    assert res["config"]["strategy"] == "fused2" and res["parity"]["ok"]
    shape, dtname, _, seed = bench.CONFIGS["c2"]
    for r in range(4):
        for g in (r * batch, (r + 1) * batch - 1):
            got = numpy.load(os.path.join(str(tmp_path), "xform_%d.npy" % g))
            ref = numpy.fft.fft(bench.global_item(shape, dtname, batch, seed, g).astype(numpy.complex128))
            assert numpy.abs(got - ref).sum() / numpy.abs(ref).sum() < EPS_F and numpy.abs(got - ref).max() <= MAX_F * numpy.abs(ref).max(), g
    assert abs(res["transforms_per_s"] - 4 * batch * res["steps"] / (res["ms_per_step"] * 1e-3 * res["steps"])) < 1e-6 * res["transforms_per_s"]

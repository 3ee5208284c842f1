"""Round 5 (GPU): BASELINE.json configs[4] as stated on one GPU -- the whole per-GPU share of 8192 transforms of 2^22 points, as the
streaming loop over 32 resident chunks and as ONE in-place execute --, eight ranks of the real sharded path, persistent
executes under stream capture, and the round's new kernels.  All through the C ABI (ctypes), against numpy.fft on the complex128-upcast
input (the reference's own oracle, test/test_errors.py:5-16,35) with the reference's thresholds."""
import json
import os
import subprocess
import sys

import numpy
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EPS_F, MAX_F = 1.1e-6, 1e-5

N5 = 1 << 22           # configuration 5: 1-D c2c fp32 N = 2^22
CHUNK5 = 256           # one resident chunk: 8 GiB in + 8 GiB out
SHARE5 = 8192          # per GPU: 65536 transforms over 8 GPUs (BASELINE.json configs[4]; kernel.py:99-121: batch -> grid)
BLK5 = 16              # the synthetic data set is periodic: 16 seeded transforms (512 MiB)

_c5_cache = {}


def _c5_block():
    """The seeded block of configuration 5 and numpy's transform of every item of it (complex128), computed once per session."""
    if "block" not in _c5_cache:
        rng = numpy.random.default_rng(1005)
        re = rng.standard_normal((BLK5, N5)).astype(numpy.float32)
        im = rng.standard_normal((BLK5, N5)).astype(numpy.float32)
        block = numpy.empty((BLK5, N5), numpy.complex64)
        block.real = re
        block.imag = im
        _c5_cache["block"] = block
        _c5_cache["refs"] = [numpy.fft.fft(block[i].astype(numpy.complex128)) for i in range(BLK5)]
    return _c5_cache["block"], _c5_cache["refs"]


def _fetch(N, ptr, item, size=N5):
    out = numpy.empty(size, numpy.complex64)
    N.check(N.lib.mifft_memcpy_d2h(out.ctypes.data, ptr + item * size * 8, out.nbytes, None))
    return out


def _close(got, ref):
    got = got.astype(numpy.complex128)
    return (numpy.abs(ref - got).sum() / numpy.abs(ref).sum() < EPS_F) and (numpy.abs(ref - got).max() <= MAX_F * numpy.abs(ref).max())


def test_config5_per_gpu_share(ctx):
    """The full per-GPU share of configuration 5 as the survey's streaming loop: 32 chunks of 256 transforms through ONE plan and
    ONE pair of 8 GiB buffers.  Global transform g holds block item (g + g // 256) % 16 -- every chunk is the block rotated one item
    further, so a chunk that was skipped, executed on stale data or mixed up with its neighbour cannot pass -- and the input buffer
    is refilled between chunks from a device copy of the block.  First / middle / last transform of EVERY chunk against numpy."""
    from pyfft_amd import _native as N
    hip = ctx.hip
    block, refs = _c5_block()
    item_bytes = N5 * 8
    dev_block = hip.to_gpu(block.reshape(-1))
    a = hip.DeviceArray((CHUNK5 * N5,), numpy.complex64)
    b = hip.DeviceArray((CHUNK5 * N5,), numpy.complex64)
    plan = hip.Plan(N5, dtype=numpy.complex64)
    assert plan.strategy(CHUNK5)[0] == "fused2"             # the persistent kernel of the stated configuration

    def refill(c):
        """a[s] <- block[(s + c) % 16] for the 256 transforms s of chunk c"""
        rot = c % BLK5
        head = (BLK5 - rot) * item_bytes
        N.check(N.lib.mifft_memcpy_d2d(a.ptr, dev_block.ptr + rot * item_bytes, head, None))
        if rot:
            N.check(N.lib.mifft_memcpy_d2d(a.ptr + head, dev_block.ptr, rot * item_bytes, None))
        done = BLK5 * item_bytes
        while done < a.nbytes:
            n = min(done, a.nbytes - done)
            N.check(N.lib.mifft_memcpy_d2d(a.ptr + done, a.ptr, n, None))
            done += n

    chunks = SHARE5 // CHUNK5
    checked = 0
    for c in range(chunks):
        refill(c)
        plan.execute(a, b, batch=CHUNK5)
        for s in (0, CHUNK5 // 2 + 1, CHUNK5 - 1):
            g = c * CHUNK5 + s
            assert _close(_fetch(N, b.ptr, s), refs[(g + g // CHUNK5) % BLK5]), ("chunk", c, "transform", g)
            checked += 1
        if c in (0, chunks // 2, chunks - 1):               # the input of an out-of-place execute stays what it was
            assert numpy.array_equal(_fetch(N, a.ptr, CHUNK5 - 1), block[(CHUNK5 - 1 + c) % BLK5])
    assert checked == 3 * chunks
    # the last chunk's result, inverse in place: the round trip
    plan.execute(b, batch=CHUNK5, inverse=True)
    for s in (0, CHUNK5 - 1):
        want = block[(s + chunks - 1) % BLK5].astype(numpy.complex128)
        got = _fetch(N, b.ptr, s).astype(numpy.complex128)
        assert numpy.abs(want - got).sum() / numpy.abs(want).sum() < EPS_F


def test_config5_share_as_one_execute(ctx):
    """The same share as ONE in-place execute(batch = 8192): 256 GiB resident (byte offsets up to 2^38, 2^35 elements, 16393 counters
    per set), skipped with a message where the allocation is refused.  Sampled transforms against numpy, periodic input ->
    bit-identical outputs across the whole buffer, inverse in place -> the input."""
    from pyfft_amd import _native as N
    hip = ctx.hip
    block, refs = _c5_block()
    try:
        buf = hip.DeviceArray((SHARE5 * N5,), numpy.complex64)
    except RuntimeError as e:
        pytest.skip("mifft_malloc refuses 256 GiB on this device: %s" % (str(e)[:200],))
    hb = block.reshape(-1).view(numpy.uint8)
    N.check(N.lib.mifft_memcpy_h2d(buf.ptr, hb.ctypes.data, hb.nbytes, None))
    done = hb.nbytes
    while done < buf.nbytes:
        n = min(done, buf.nbytes - done)
        N.check(N.lib.mifft_memcpy_d2d(buf.ptr + done, buf.ptr, n, None))
        done += n
    N.check(N.lib.mifft_device_sync())
    plan = hip.Plan(N5, dtype=numpy.complex64)
    assert plan.strategy(SHARE5)[0] == "fused2"
    plan.execute(buf, batch=SHARE5)
    samples = [0, 1, BLK5 - 1, BLK5, SHARE5 // 2 - 1, SHARE5 // 2, SHARE5 // 2 + 5, SHARE5 - BLK5 - 3, SHARE5 - 2, SHARE5 - 1]
    first = {}
    for g in samples:
        got = _fetch(N, buf.ptr, g)
        assert _close(got, refs[g % BLK5]), g
        # periodic input -> bit-identical output wherever the item lies in the 256 GiB
        if g % BLK5 in first:
            assert numpy.array_equal(first[g % BLK5].view(numpy.uint32), got.view(numpy.uint32)), g
        else:
            first[g % BLK5] = got
    plan.execute(buf, batch=SHARE5, inverse=True)
    for g in (0, SHARE5 // 2 + 5, SHARE5 - 1):
        want = block[g % BLK5].astype(numpy.complex128)
        got = _fetch(N, buf.ptr, g).astype(numpy.complex128)
        assert numpy.abs(want - got).sum() / numpy.abs(want).sum() < EPS_F, g
    plan.close()
    del buf


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


# ---- persistent executes under stream capture / hipGraph replay -----------------------------------------------------------------
def _tiled_noise(count, dtype, seed):
    rng = numpy.random.default_rng(seed)
    cdt = numpy.dtype(dtype)
    fdt = numpy.float32 if cdt == numpy.complex64 else numpy.float64
    out = numpy.empty(count, cdt)
    for part in ("real", "imag"):
        blk = rng.standard_normal(min(count, (1 << 22) + 17)).astype(fdt)
        setattr(out, part, numpy.resize(blk, count))
    return out


CAPTURE_CASES = [((1 << 18,), 160, numpy.complex64, "fused2"),        # 28 / 56 ring: the review's case
                 ((1 << 20,), 64, numpy.complex64, "fused2"),         # BASELINE config 2's kernel
                 ((128, 128, 128), 32, numpy.complex64, "fusedp"),
                 ((1024, 1024), 64, numpy.complex64, "fused2"),       # the 2-D form
                 ((1 << 16,), 96, numpy.complex64, "chain"),
                 ((512, 1024), 80, numpy.complex128, "fused2"),
                 ((256, 4096), 40, numpy.complex64, "pipelined")]     # cache-sized chunks: eager on the plan's side streams, recorded as a linear graph


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
    # a batch the plan has not run yet cannot be captured (its scratch would be allocated inside the capture): loud, not wrong
    with pytest.raises(RuntimeError, match="eager execute"):
        with hip.Graph(s):
            plan.execute(a, b, batch=batch - 1)
    assert not plan._context.capturing()                # (the failed body left no capture behind)
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


def test_direct_abi_two_set_launch_refuses_capture_and_null_error_word(ctx):
    """C-ABI users of mifft_launch_fused2: the two-set form on a capturing stream is MIFFT_E_INVALID (a replay would start on dirty
    counters), and so is the two-set form without an error word of its own (the next launch would zero the default one)."""
    import ctypes
    from pyfft_amd import _native as N
    hip = ctx.hip
    n, batch = 1 << 18, 160
    s = hip.Stream()
    plan = hip.Plan(n, dtype=numpy.complex64, stream=s)
    a = hip.DeviceArray((n * batch,), numpy.complex64)
    N.check(N.lib.mifft_memset(a.ptr, 0, a.nbytes, s.handle))
    plan.execute(a, batch=batch)
    s.synchronize()
    strat = plan.strategy(batch)
    assert strat[0] == "fused2"
    _, lag, ring, grid = strat
    descs = plan._descriptors(batch, True, False)
    base = plan._context.pointer_of(plan._counters)
    nb = plan._counter_bytes
    N.check(N.lib.mifft_memset(base, 0, 3 * nb, s.handle))
    s.synchronize()
    plan._counters_clean, plan._counter_set = True, 0
    tmp = plan._context.pointer_of(plan._tempmemobj)

    def launch(sync):
        return N.lib.mifft_launch_fused2(ctypes.byref(descs[0]), ctypes.byref(descs[1]), a.ptr, None, a.ptr, None, tmp, None, ring, lag,
                                         ctypes.byref(sync), grid, s.handle)

    assert launch(N.MifftFusedSync(base, base + nb, None)) == N.E_INVALID
    assert b"error word" in N.lib.mifft_last_error()
    N.check(N.lib.mifft_stream_begin_capture(s.handle))
    rc = launch(N.MifftFusedSync(base, base + nb, plan._errword.ptr))
    msg = N.lib.mifft_last_error()
    rc_single = launch(N.MifftFusedSync(base + 2 * nb, None, plan._errword.ptr))
    h = ctypes.c_void_p()
    N.check(N.lib.mifft_stream_end_capture(s.handle, ctypes.byref(h)))
    N.lib.mifft_graph_destroy(h)
    assert rc == N.E_INVALID and b"capturing" in msg
    assert rc_single == 0
    s.synchronize()


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


# ---- one pass pair instead of a third launch (csrc/fft_pair_f32.hip / _f64.hip, pyfft_amd/passes.py) ---------------------------------
PAIR_CHAIN_CASES = [((4096, 256), numpy.complex64, 2), ((4096, 256), numpy.complex128, 1),
                    ((32, 32, 2048), numpy.complex64, 2), ((16, 16, 2048), numpy.complex64, 5),
                    ((32, 32, 1024), numpy.complex128, 2), ((16, 16, 1024), numpy.complex128, 5), ((2, 4096, 256), numpy.complex64, 1)]
_PAIR_CHAIN_SOAK = [((4096, 512), numpy.complex64, 1), ((4096, 512), numpy.complex128, 1), ((32, 32, 4096), numpy.complex64, 1),
                    ((16, 16, 4096), numpy.complex64, 1), ((32, 32, 2048), numpy.complex128, 1), ((16, 16, 2048), numpy.complex128, 2)]
PAIR_CHAIN_CASES += [((4096, 128), numpy.complex128, 3), ((4096, 128), numpy.complex64, 5)]      # (late in round 5: 128-point rows)
# every one of these is a pair-kernel instance of its own (csrc/fft_pair_f32.hip / _f64.hip) that a default plan selects: all in the default suite
# (round 6; the biggest planes are 16 ... 256 MiB per transform: numpy takes a second or two per case)
PAIR_CHAIN_CASES += _PAIR_CHAIN_SOAK + [((4096, 4096), numpy.complex128, 1),
                                        ((4096, 1024), numpy.complex64, 3), ((4096, 2048), numpy.complex64, 1), ((4096, 4096), numpy.complex64, 1),
                                        ((4096, 1024), numpy.complex128, 2), ((4096, 2048), numpy.complex128, 1)]


@pytest.mark.parametrize("shape,dtype,batch", PAIR_CHAIN_CASES, ids=lambda v: getattr(v, "__name__", str(v)).replace(" ", ""))
def test_pair_chains_two_launches(ctx, shape, dtype, batch):
    """Shapes that ran THREE launches through round 4 -- a 2-D shape with a 4096-point y axis (row + two strided passes: the reference's
    own factorisation of a long axis, pyfft/kernel.py:259-283) and a 3-D shape with short y and z behind a long x (one chain per axis,
    pyfft/plan.py:160-167) -- now run one pass PAIR and one single pass.  The reference's six-assertion protocol against numpy
    (test/test_errors.py:18-114: out of place, in place, forward, inverse, input untouched) at ragged batches, and the chain really has
    two launches."""
    from test_errors_gpu import run_protocol
    from pyfft_amd.passes import launch_units
    plan = ctx.getPlan(shape, dtype=dtype)
    units = launch_units(plan.pass_list())
    assert len(units) == (3 if len(shape) == 3 and shape[0] == 2 else 2), plan.pass_list()
    assert sum(1 for k in plan.pass_list() if k.pair_with_next) == 1
    run_protocol(ctx, shape, dtype, batch, seed=5150 + batch)


@pytest.mark.parametrize("shape,dtype,batch", [((4096, 256), numpy.complex64, 320), ((32, 32, 2048), numpy.complex64, 160), ((4096, 512), numpy.complex128, 96)],
                         ids=lambda v: getattr(v, "__name__", str(v)).replace(" ", ""))
def test_pair_chains_pipelined_chunks(ctx, shape, dtype, batch):
    """The same chains beyond the chain threshold: cache-sized chunks on two streams, a pair launch + a single launch per chunk.
    The whole array against the plain chain's (same kernels: identical bits), sampled transforms against numpy."""
    hip = ctx.hip
    size = int(numpy.prod(shape))
    cdt = numpy.dtype(dtype)
    data = _tiled_noise(size * batch, dtype, 777)
    tol, tol_max = (1.1e-6, 1e-5) if cdt == numpy.complex64 else (1e-11, 1e-10)
    outs = {}
    for strat in ("chain", "auto"):
        os.environ["PYFFT_AMD_STRATEGY"] = strat
        try:
            plan = hip.Plan(shape, dtype=dtype)
            st = plan.strategy(batch)[0]
            assert st == ("chain" if strat == "chain" else "pipelined"), st
            a, b = hip.to_gpu(data), hip.DeviceArray((size * batch,), dtype)
            plan.execute(a, b, batch=batch)
            outs[strat] = b.get()
        finally:
            os.environ.pop("PYFFT_AMD_STRATEGY", None)
    assert numpy.array_equal(outs["chain"].view(numpy.uint8), outs["auto"].view(numpy.uint8))
    for item in (0, batch // 2, batch - 1):
        ref = numpy.fft.fftn(data[item * size:(item + 1) * size].astype(numpy.complex128).reshape(shape)).reshape(-1)
        got = outs["auto"][item * size:(item + 1) * size].astype(numpy.complex128)
        assert numpy.abs(ref - got).sum() / numpy.abs(ref).sum() < tol and numpy.abs(ref - got).max() <= tol_max * numpy.abs(ref).max()


# ---- one-tile-per-CU N-D shapes as two work-groups per transform (csrc/fft_nd2z.hpp) ------------------------------------------------
ND2Z_SHAPES = {numpy.complex64: [(1024, 32), (512, 64), (256, 128), (128, 256), (32, 1024), (8, 64, 64), (16, 16, 128), (32, 32, 32),
                                 # (two-per-CU shapes: split in small launches only -- 11 transforms are one)
                                 (1024, 16), (512, 32), (256, 64), (128, 128), (64, 256), (32, 512), (16, 1024), (16, 32, 32), (32, 16, 32), (32, 32, 16)],
               numpy.complex128: [(512, 32), (256, 64), (64, 256), (32, 512), (16, 32, 32), (16, 16, 64), (128, 128), (64, 16, 16),
                                  (512, 16), (256, 32), (128, 64), (64, 128), (32, 256), (16, 512), (16, 16, 32), (16, 32, 16), (32, 16, 16)]}


@pytest.mark.parametrize("dtype", [numpy.complex64, numpy.complex128], ids=["c64", "c128"])
def test_two_work_groups_per_transform_nd(ctx, dtype):
    """The N-D shapes of 32768 points (fp32) / 16384 points (fp64) as TWO work-groups per transform (one decimation-in-frequency step
    along the slowest axis folded into the loads, each half on a two-per-CU tile; numpy shapes = (z, y, x)): every such shape at a
    ragged batch against numpy with the reference's thresholds (test/test_errors.py:20-23), forward and inverse, against the
    one-tile-per-CU kernel to rounding, and an in-place call -- which must take the one-tile kernel -- bit-identical to it."""
    from pyfft_amd import _native as N
    hip = ctx.hip
    cdt = numpy.dtype(dtype)
    tol, tol_max, tol_same = (1.1e-6, 1e-5, 5e-7) if cdt == numpy.complex64 else (1e-11, 1e-10, 1e-14)
    for shape in ND2Z_SHAPES[dtype]:
        size = int(numpy.prod(shape))
        batch = 11
        data = _tiled_noise(size * batch, dtype, 4400 + shape[0])
        plan = hip.Plan(shape, dtype=dtype)
        assert len(plan.pass_list()) == 1 and plan.pass_list()[0].kind == N.PASS_ND, (shape, plan.pass_list())
        a = hip.to_gpu(data)
        outs = {}
        for alt in (6, 5):                       # 6: the one-tile-per-CU kernel, 5 (= the default): two work-groups per transform
            N.check(N.lib.mifft_debug_set(N.DEBUG_ALT_ROWS, alt), "debug_set")
            try:
                b = hip.DeviceArray((size * batch,), dtype)
                plan.execute(a, b, batch=batch)
                outs[alt] = b.get()
                if alt == 5:
                    plan.execute(b, batch=batch, inverse=True)          # in place: the one-tile kernel whatever the switch says
                    back = b.get()
                    c = hip.DeviceArray((size * batch,), dtype)
                    plan.execute(hip.to_gpu(outs[5]), c, batch=batch, inverse=True)     # out of place: the two-work-group form
                    back2 = c.get()
            finally:
                N.check(N.lib.mifft_debug_set(N.DEBUG_ALT_ROWS, 0), "debug_set")
        assert numpy.array_equal(a.get(), data), "input modified"
        for item in range(batch):
            sl = slice(item * size, (item + 1) * size)
            ref = numpy.fft.fftn(data[sl].astype(numpy.complex128).reshape(shape)).reshape(-1)
            got = outs[5][sl].astype(numpy.complex128)
            assert numpy.abs(ref - got).sum() / numpy.abs(ref).sum() < tol, (shape, item)
            assert numpy.abs(ref - got).max() <= tol_max * numpy.abs(ref).max(), (shape, item)
        d = numpy.abs(outs[5].astype(numpy.complex128) - outs[6]).sum() / numpy.abs(outs[6]).sum()
        assert d < tol_same, (shape, d)
        for inv in (back, back2):
            assert numpy.abs(inv.astype(numpy.complex128) - data).sum() / numpy.abs(data).sum() < tol, shape


OOP_ND_CASES = [(sh, numpy.complex64) for sh in [(256, 256), (512, 128), (1024, 64), (64, 1024), (16, 64, 64),
                                                  # 32768-point shapes without a one-tile kernel: two work-groups per transform
                                                  (64, 512), (16, 2048), (2048, 16), (64, 8, 64), (16, 128, 16), (128, 16, 16), (32, 16, 64), (32, 64, 16)]] + \
               [(sh, numpy.complex128) for sh in [(16, 1024), (1024, 16), (8, 32, 64), (64, 8, 32), (4, 64, 64), (64, 4, 64), (32, 16, 32), (16, 64, 16), (32, 32, 16)]]


@pytest.mark.parametrize("shape,dtype", OOP_ND_CASES, ids=lambda v: getattr(v, "__name__", "x".join(map(str, v)) if isinstance(v, tuple) else str(v)))
def test_four_work_groups_per_transform_out_of_place(ctx, shape, dtype, monkeypatch):
    """Shapes with a one-launch kernel for OUT-OF-PLACE executes only (csrc/fft_nd2z.hpp; two launches as a chain, which the plan keeps for
    in-place executes): 65536 points (fp32) on four work-groups per transform, and the 32768-point (fp32) / 16384-point (fp64) shapes that
    have no one-tile kernel on two.  The reference's thresholds against numpy at a ragged batch, forward and inverse, out of place and in
    place, and the chain's result to rounding."""
    hip = ctx.hip
    size = int(numpy.prod(shape))
    csz = numpy.dtype(dtype).itemsize
    batch = (261 * 65536 * 8) // (size * csz)                     # 130.5 MiB per side: beyond half the cache, where the plan uses every such kernel
    EPS, MAXN, SAME = (EPS_F, MAX_F, 5e-7) if numpy.dtype(dtype) == numpy.complex64 else (1e-11, 1e-10, 1e-14)
    data = _tiled_noise(size * batch, dtype, 4500 + shape[0])
    plan = hip.Plan(shape, dtype=dtype)
    assert plan._oop_nd is not None and len(plan.pass_list()) == 2, plan.pass_list()
    assert batch * size * csz > plan._context.machine.write_through_max_bytes
    a = hip.to_gpu(data)
    b = hip.DeviceArray((size * batch,), dtype)
    plan.execute(a, b, batch=batch)
    got = b.get()
    assert numpy.array_equal(a.get(), data), "input modified"
    for item in (0, 1, 7, 8, 9, batch // 2, batch - 6, batch - 5, batch - 1):      # (groups of eight transforms share an XCD: both ends of the last, ragged group)
        sl = slice(item * size, (item + 1) * size)
        ref = numpy.fft.fftn(data[sl].astype(numpy.complex128).reshape(shape)).reshape(-1)
        g = got[sl].astype(numpy.complex128)
        assert numpy.abs(ref - g).sum() / numpy.abs(ref).sum() < EPS, (shape, item)
        assert numpy.abs(ref - g).max() <= MAXN * numpy.abs(ref).max(), (shape, item)
    c = hip.to_gpu(data)
    plan.execute(c, batch=batch)                                  # in place: the chain
    inplace = c.get()
    assert numpy.abs(inplace.astype(numpy.complex128) - got).sum() / numpy.abs(got).sum() < SAME
    monkeypatch.setenv("PYFFT_AMD_NO_OOP_ND", "1")
    d = hip.DeviceArray((size * batch,), dtype)
    hip.Plan(shape, dtype=dtype).execute(a, d, batch=batch)       # out of place on the chain
    assert numpy.array_equal(d.get(), inplace)
    monkeypatch.delenv("PYFFT_AMD_NO_OOP_ND")
    plan.execute(b, a, batch=batch, inverse=True)                 # inverse, out of place
    assert numpy.abs(a.get().astype(numpy.complex128) - data).sum() / numpy.abs(data).sum() < EPS


# ---- tiny one-launch N-D shapes routed to the run-time-shaped kernel (tuning table "nd_generic") -------------------------------------------
@pytest.mark.parametrize("shape,dtype,batch", [((16, 2), numpy.complex64, 37), ((2, 8), numpy.complex64, 100), ((4, 4), numpy.complex128, 61)],
                         ids=lambda v: getattr(v, "__name__", str(v)).replace(" ", ""))
def test_tiny_nd_shapes_small_launches(ctx, monkeypatch, shape, dtype, batch):
    """Shapes of the tuning table's "nd_generic" lists in small launches (the "always" shape on the run-time-shaped kernel, the "big" ones on
    their fixed instances): the reference's six-assertion protocol against numpy (test/test_errors.py:18-114), either way round."""
    from test_errors_gpu import run_protocol
    run_protocol(ctx, shape, dtype, batch, seed=977)
    monkeypatch.setenv("PYFFT_AMD_NO_ND_GENERIC", "1")
    run_protocol(ctx, shape, dtype, batch, seed=977, check_oracle=False)


@pytest.mark.parametrize("shape,dtype,batch", [((16, 2), numpy.complex64, 700001), ((8, 8), numpy.complex64, 270001), ((4, 4), numpy.complex128, 530001),
                                               ((4, 2), numpy.complex64, 2100001), ((2, 8), numpy.complex64, 1100001), ((8, 2), numpy.complex128, 600001)],
                         ids=lambda v: getattr(v, "__name__", str(v)).replace(" ", ""))
def test_tiny_nd_shapes_big_launches(ctx, monkeypatch, shape, dtype, batch):
    """The same shapes in launches beyond the size rule (130-270 MiB per side, ragged batches), where the plan marks the pass variant 1: the
    run-time-shaped kernel's result against the fixed instance's over the WHOLE array (same transform, another operation order: the
    reference's L1 threshold and the north star's max-norm bound), 64 sampled transforms against numpy, in place == out of place, the
    input untouched, and the inverse round trip."""
    hip = ctx.hip
    size = int(numpy.prod(shape))
    cdt = numpy.dtype(dtype)
    eps, mx = (1.1e-6, 1e-5) if cdt == numpy.complex64 else (1e-11, 1e-11)
    data = _tiled_noise(size * batch, dtype, 611)
    plan = hip.Plan(shape, dtype=dtype, wait_for_finish=True)
    assert plan._descriptors(batch, False, False)[0].variant == 1
    a, b = hip.to_gpu(data), hip.DeviceArray((size * batch,), dtype)
    plan.execute(a, b, batch=batch)
    got = b.get()
    assert numpy.array_equal(a.get(), data), "an out-of-place execute touched its input"
    c = hip.to_gpu(data)
    plan.execute(c, batch=batch)
    assert numpy.array_equal(c.get(), got), "in place differs from out of place"
    plan.execute(c, batch=batch, inverse=True)
    back = c.get()
    assert numpy.abs(back - data).sum() / numpy.abs(data).sum() < eps
    for item in numpy.linspace(0, batch - 1, 64).astype(int):
        ref = numpy.fft.fftn(data[item * size:(item + 1) * size].reshape(shape).astype(numpy.complex128)).reshape(-1)
        g = got[item * size:(item + 1) * size]
        assert numpy.abs(ref - g).sum() / numpy.abs(ref).sum() < eps and numpy.abs(ref - g).max() <= max(mx, 1e-5 if cdt == numpy.complex64 else 1e-11) * numpy.abs(ref).max(), item
    monkeypatch.setenv("PYFFT_AMD_NO_ND_GENERIC", "1")
    fixed = hip.Plan(shape, dtype=dtype, wait_for_finish=True)
    assert fixed._descriptors(batch, False, False)[0].variant == 0
    fixed.execute(a, b, batch=batch)
    want = b.get()
    assert numpy.abs(want - got).sum() / numpy.abs(want).sum() < eps
    assert numpy.abs(want - got).max() <= 1e-5 * numpy.abs(want).max()


# ---- 3-D shapes with 256-point rows next to a shorter axis: the two pass pairs of 256^3 instead of three launches ---------------------------
_LATE_PAIR_SHAPES = [((64, 128, 256), numpy.complex128, 1), ((64, 256, 256), numpy.complex64, 1), ((32, 256, 128), numpy.complex128, 3)]
if True:      # (round 6: every shape is a pair-kernel instance of its own -- all in the default suite)
    _LATE_PAIR_SHAPES += [((64, 256, 128), numpy.complex128, 1), ((128, 256, 128), numpy.complex128, 1), ((256, 256, 128), numpy.complex128, 1),
                          ((32, 256, 256), numpy.complex128, 1), ((32, 128, 256), numpy.complex128, 2),
                          ((128, 256, 64), numpy.complex128, 1), ((32, 256, 64), numpy.complex128, 5), ((256, 256, 64), numpy.complex128, 1),
                          ((64, 256, 64), numpy.complex128, 2), ((32, 256, 256), numpy.complex64, 3),
                          ((128, 256, 256), numpy.complex128, 1), ((64, 256, 256), numpy.complex128, 1), ((128, 128, 256), numpy.complex128, 2),
                          ((256, 128, 256), numpy.complex128, 1), ((128, 256, 256), numpy.complex64, 1)]


@pytest.mark.parametrize("shape,dtype,batch", _LATE_PAIR_SHAPES, ids=lambda v: getattr(v, "__name__", str(v)).replace(" ", ""))
def test_pass_pairs_for_256_point_rows(ctx, monkeypatch, shape, dtype, batch):
    """(z, y, 256) with y in {128, 256}, z in {32 ... 256}, and (z, 256, 128) in complex128; (z, 256, 256), z in {64, 128}, in complex64: (ROW x, COL y R0) and
    (COL y R1, COL z) as two launches (csrc/fft_pair_f64.hip: y = 32 x 8 / 32 x 4; fft_pair_f32.hip: 64 x 4; pyfft/kernel.py:259-283
    splits a long axis the same way).  The reference's six-assertion protocol against numpy, and the same data through the one-pass-per-axis chain (pairs
    switched off) within the same thresholds."""
    from test_errors_gpu import run_protocol
    from pyfft_amd import _native as N
    from pyfft_amd.passes import launch_units
    plan = ctx.getPlan(shape, dtype=dtype)
    assert len(launch_units(plan.pass_list())) == 2 and sum(1 for k in plan.pass_list() if k.pair_with_next) == 2, plan.pass_list()
    run_protocol(ctx, shape, dtype, batch, seed=8100 + shape[0])
    N.lib.mifft_debug_set(N.DEBUG_PAIR, 1)
    try:
        chain = ctx.getPlan(shape, dtype=dtype)
        # (three launches, or two where the (y, x) plane has a one-tile kernel: (z, 256, 64) in fp64 -- the pairs measured 0.29 -> 0.39 there)
        assert len(launch_units(chain.pass_list())) in (2, 3) and not any(k.pair_with_next for k in chain.pass_list())
        run_protocol(ctx, shape, dtype, batch, seed=8100 + shape[0], check_oracle=False)
    finally:
        N.lib.mifft_debug_set(N.DEBUG_PAIR, 0)

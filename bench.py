#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X batched FFT hot path.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): batched 1-D c2c fp32, N = 2^20, batch 4096 per GPU, out of
place, synthetic N(0,1) data resident in HBM before the timed region.  One "step" = one
plan.execute() over the whole per-GPU batch (every pass of the transform).

N > 1: one process per GPU; the batch is sharded across ranks and the hot path has no collective
(torch.distributed / RCCL only for the barrier and the max-over-ranks reduction), weak scaling.
Under a launcher (RANK / WORLD_SIZE set, e.g. `python -m torch.distributed.run ...`) this process is
one rank.  Without one, `--gpus N` with N > 1 starts the N ranks itself, as children, BEFORE
anything in this process touches torch or HIP, and relays rank 0's line.

Prints ONE JSON line on rank 0.  `value` = nominal 5*N*log2(N) GFLOPS of the whole job over EXACTLY K
timed steps (test/test_performance.py:24 in the reference); the line also carries transforms/s,
the algorithmic HBM GB/s, the roofline object (HIP events on the plan's stream), the reference's
timing protocol (out of place AND in place, median of >= 5 repeats: test/test_performance.py:22-30,
cuda/test.cu:37-64) and the CPU baseline.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md

CONFIGS = {
    # name: (shape, dtype name, per-GPU batch, seed)            BASELINE.json configs[i]
    "c1": ((1024,), "complex64", 1, 1001),
    "c2": ((1 << 20,), "complex64", 4096, 1002),
    "c3": ((1024, 1024), "complex64", 512, 1003),
    "c4": ((256, 256, 256), "complex128", 64, 1004),
    "c4s": ((256, 256, 256), "float64", 64, 1004),
    "c5": ((1 << 22,), "complex64", 256, 1005),   # per-GPU resident chunk of config 5 (8 GiB in + 8 GiB out)
    # not BASELINE configurations: the reference's published 128^3 row (doc/source/index.rst:373) at 1 GiB per side, for
    # tools/pmc_traffic.py and the round-4 evidence of the persistent two-pair kernel
    "cube": ((128, 128, 128), "complex64", 64, 1006),
    "cubed": ((128, 128, 128), "complex128", 32, 1007),
    # configs 2 and 3 in the reference's OTHER layout (dtype float32: re / im planes), 4 GiB per side
    "c2s": ((1 << 20,), "float32", 512, 1008),
    "c3s": ((1024, 1024), "float32", 512, 1009),
}


C5_SHARE = 8192     # transforms per GPU of BASELINE.json configs[4] (65536 over 8 GPUs)


def shard_batch(global_batch, rank, world):
    """Contiguous slice [start, start+count) of the batch axis owned by `rank`: independent transforms, no
    exchange.  The library's own rule (pyfft_amd/sharded.py: the split ShardedPlan makes); imported lazily so that the parent of a
    self-launched job loads nothing."""
    from pyfft_amd.sharded import shard_batch as _sb
    return _sb(global_batch, rank, world)


def dist_env():
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    return rank, local_rank, world


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def visible_gpus():
    """Number of GPUs this process may use, found WITHOUT the HIP runtime: the *_VISIBLE_DEVICES lists if set, else the KFD
    topology nodes that have SIMDs.  None if neither source exists."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    base = "/sys/class/kfd/kfd/topology/nodes"
    if not os.path.isdir(base):
        return None
    n = 0
    for node in os.listdir(base):
        try:
            with open(os.path.join(base, node, "properties")) as f:
                for line in f:
                    k, _, v = line.partition(" ")
                    if k == "simd_count" and int(v) > 0:
                        n += 1
        except (OSError, ValueError):
            pass
    return n


def self_launch(args, argv):
    """`--gpus N` without a launcher: start the N ranks as child processes (torch.distributed.run, rendezvous on
    127.0.0.1) and relay their output.  Called before this process imports torch or touches HIP: a process that has
    initialised the GPU must never be replaced or forked into another program."""
    n = args.gpus
    note = None
    if not args.selftest_dist and not args.share_gpu:
        # The parent only SPAWNS children: it counts the GPUs without touching torch or HIP (sysfs), so that nothing here can
        # initialise the runtime; the children do the real work.  Unknown count (no sysfs): the children fail loudly.
        visible = visible_gpus()
        if visible is not None and visible < 1:
            raise SystemExit("bench.py: no GPU visible")
        if visible is not None and visible < n:
            note = "requested %d GPUs, %d visible: ran %d ranks" % (n, visible, visible)
            n = visible
    child = [a for a in argv]
    for i, a in enumerate(child):           # the children see the number of ranks that really run
        if a == "--gpus" and i + 1 < len(child):
            child[i + 1] = str(n)
        elif a.startswith("--gpus="):
            child[i] = "--gpus=%d" % n
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + child
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout.splitlines():
        if line.startswith("{") and note:
            try:
                d = json.loads(line)
                d["note"] = note
                line = json.dumps(d)
            except ValueError:
                pass
        print(line)
    raise SystemExit(proc.returncode)


def selftest_dist(args):
    """CPU self-test of the multi-process harness (gloo): sharding, barrier, max-over-ranks.
    No FFT is computed and no benchmark number is produced."""
    import torch
    import torch.distributed as dist
    rank, local_rank, world = dist_env()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gb = 4096 * world + 3
    start, count = shard_batch(gb, rank, world)
    t = torch.tensor([float(count), float(start)])
    gathered = [torch.zeros(2) for _ in range(world)]
    dist.all_gather(gathered, t)
    dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))          # stand-in for K steps; rank-dependent on purpose
    dist.barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    mx = el.clone()
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    if rank == 0:
        counts = [int(g[0]) for g in gathered]
        starts = [int(g[1]) for g in gathered]
        ok = sum(counts) == gb and starts == [sum(counts[:i]) for i in range(world)] and mx.item() >= el.item()
        print(json.dumps({"selftest": True, "world": world, "counts": counts, "starts": starts,
                          "max_s": mx.item(), "ok": bool(ok)}))
    dist.destroy_process_group()


def fill_device(N, dst_ptr, nbytes, host_block):
    """Upload one host block and tile it across the buffer with device-to-device copies."""
    hb = host_block.view("uint8").reshape(-1)
    n0 = min(nbytes, hb.nbytes)
    N.check(N.lib.mifft_memcpy_h2d(dst_ptr, hb.ctypes.data, n0, None), "h2d")
    done = n0
    while done < nbytes:
        n = min(done, nbytes - done)
        N.check(N.lib.mifft_memcpy_d2d(dst_ptr + done, dst_ptr, n, None), "d2d")
        done += n
    N.check(N.lib.mifft_device_sync(), "sync")


_POOL_ITEMS = None


def _pool_fft(i):
    import numpy
    numpy.fft.fftn(_POOL_ITEMS[i % len(_POOL_ITEMS)])
    return 0


def cpu_baseline(shape, dtype, host_items, flop_per_xform, budget_s=24.0, max_workers=0):
    """CPU figures on this box's host cores for the same transforms.  Runs BEFORE this process
    initialises the GPU (the process pool forks).  host_items: array [n, *shape] of the very data the GPU transforms.
    The reference has no CPU implementation of its own: its CPU path is numpy.fft (test/test_errors.py:5-16,35), which is
    therefore the `value` (kind "reference"), fanned over all cores with a process pool; scipy.fft with all cores and the
    oracle's plain-C restatement of the reference chain (a scalar port, one core) are reported next to it."""
    import numpy
    global _POOL_ITEMS
    res = {}
    axes = tuple(range(1, host_items.ndim))
    cores = host_cores = os.cpu_count() or 1
    if max_workers > 0:                 # (tests: a pool of 256 processes costs ten seconds to start and stop whatever the budget)
        cores = min(cores, max_workers)
    n = host_items.shape[0]
    part = budget_s / 4.0
    # (1) numpy.fft (pocketfft, one thread)
    t0 = time.perf_counter()
    done = 0
    for i in range(n):
        numpy.fft.fftn(host_items[i])
        done += 1
        if time.perf_counter() - t0 > part:
            break
    dt = time.perf_counter() - t0
    rate1 = done / dt
    res["numpy_fft_1core_gflops"] = flop_per_xform * rate1 / 1e9
    res["numpy_fft_1core_xforms_per_s"] = rate1
    res["numpy_sample_xforms"] = done
    # (2) numpy.fft over a process pool of all cores
    pool_val = None
    try:
        import multiprocessing as mp
        _POOL_ITEMS = host_items
        workers = cores
        tasks = max(workers, min(4 * workers, int(rate1 * workers * part * 0.5) or workers))
        with mp.get_context("fork").Pool(workers) as pool:
            pool.map(_pool_fft, range(workers), chunksize=1)          # start-up and first touch, untimed
            t0 = time.perf_counter()
            pool.map(_pool_fft, range(tasks), chunksize=1)
            dt = time.perf_counter() - t0
        pool_val = flop_per_xform * tasks / dt / 1e9
        res["numpy_fft_pool_gflops"] = pool_val
        res["numpy_fft_pool_xforms_per_s"] = tasks / dt
        res["numpy_fft_pool_workers"] = workers
        res["numpy_fft_pool_sample_xforms"] = tasks
    except Exception as e:
        res["numpy_pool_error"] = repr(e)
    finally:
        _POOL_ITEMS = None
    # (3) scipy.fft with all cores (a stronger CPU figure than the reference's own path)
    try:
        import scipy.fft
        scipy.fft.fftn(host_items[:1], axes=axes, workers=cores)
        t0 = time.perf_counter()
        reps = 0
        while True:
            scipy.fft.fftn(host_items, axes=axes, workers=cores)
            reps += 1
            if time.perf_counter() - t0 > part or reps >= 8:
                break
        dt = time.perf_counter() - t0
        res["scipy_fft_allcores_gflops"] = flop_per_xform * n * reps / dt / 1e9
        res["scipy_fft_allcores_xforms_per_s"] = n * reps / dt
    except Exception as e:  # scipy missing
        res["scipy_error"] = repr(e)
    # (4) the oracle's plain-C restatement of the reference chain (scalar port, one core)
    try:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import c_oracle
        import pyfft_oracle
        if c_oracle.available():
            _, xyz = pyfft_oracle.normalize_shape(tuple(shape))
            t0 = time.perf_counter()
            k = 0
            while k < n:
                c_oracle.execute(host_items[k].reshape(-1), xyz, batch=1)
                k += 1
                if time.perf_counter() - t0 > part:
                    break
            dt = time.perf_counter() - t0
            res["oracle_c_port_1core_gflops"] = flop_per_xform * k / dt / 1e9
            res["oracle_c_port_sample_xforms"] = k
    except Exception as e:
        res["oracle_port_error"] = repr(e)
    if pool_val is not None:
        value, used, impl, sample = pool_val, cores, "numpy.fft.fftn, process pool of %d workers" % cores, res["numpy_fft_pool_sample_xforms"]
    else:
        value, used, impl, sample = res["numpy_fft_1core_gflops"], 1, "numpy.fft.fftn (1 thread)", done
    # the strongest CPU figure of the run, whatever library it comes from (the reference's own path is `value`)
    cands = [(res.get("numpy_fft_pool_gflops"), "numpy.fft.fftn, process pool of %d workers" % cores, cores),
             (res.get("numpy_fft_1core_gflops"), "numpy.fft.fftn (1 thread)", 1),
             (res.get("scipy_fft_allcores_gflops"), "scipy.fft.fftn(workers=%d)" % cores, cores)]
    bv, bimpl, bcores = max((c for c in cands if c[0] is not None), key=lambda c: c[0])
    best = {"value": bv, "unit": "GFLOPS", "impl": bimpl, "cores": bcores}
    out = {"value": value, "unit": "GFLOPS", "cores": used, "kind": "reference", "best": best,
           "impl": impl + " -- the reference's own CPU path (test/test_errors.py:35)",
           "sample": "%d transforms drawn from %d items of the same %s %s data the GPU transforms (host copy), nominal 5*N*log2(N) flop" %
                     (sample, n, "x".join(map(str, shape)), dtype),
           "host_cores": host_cores}
    out.update(res)
    return out


def make_host_block(shape, dtname, batch, seed, gstart=0):
    """The GLOBAL synthetic dataset is periodic in the transform index: transform g of the job holds block item g % blk of ONE
    seeded block (the same on every rank).  A rank whose slice starts at global index `gstart` gets the block rotated so that
    its local transform s holds global transform gstart + s -- `host_c[s % blk]` is then the data of local transform s, and a
    rank's output can be checked against numpy on the global dataset's slice [gstart, gstart + count)."""
    import numpy
    dtype = numpy.dtype(dtname)
    cdtype = numpy.dtype(numpy.complex64 if dtype in (numpy.complex64, numpy.float32) else numpy.complex128)
    size = int(numpy.prod(shape))
    blk = min(batch, max(1, (512 << 20) // (size * cdtype.itemsize)), 64)
    rng = numpy.random.default_rng(seed)
    fdt = numpy.float32 if cdtype == numpy.complex64 else numpy.float64
    host_re = rng.standard_normal((blk,) + tuple(shape)).astype(fdt)
    host_im = rng.standard_normal((blk,) + tuple(shape)).astype(fdt)
    if gstart % blk:
        host_re = numpy.roll(host_re, -(gstart % blk), axis=0)
        host_im = numpy.roll(host_im, -(gstart % blk), axis=0)
    host_c = numpy.empty((blk,) + tuple(shape), cdtype)
    host_c.real = host_re
    host_c.imag = host_im
    return blk, host_re, host_im, host_c


def global_item(shape, dtname, batch, seed, g):
    """Transform g of the global dataset, regenerated independently of any rank (test helper)."""
    blk, _, _, host_c = make_host_block(shape, dtname, batch, seed, 0)
    return host_c[g % blk]


def library_digest():
    """First 16 hex digits of the sha256 of the libmifft.so this process runs (what profiles/traffic_<c>.json records)."""
    import hashlib
    try:
        with open(os.path.join(ROOT, "pyfft_amd", "libmifft.so"), "rb") as f:
            return hashlib.sha256(f.read()).hexdigest()[:16]
    except OSError:
        return None


def stats(xs):
    xs = sorted(xs)
    n = len(xs)
    med = xs[n // 2] if n % 2 else 0.5 * (xs[n // 2 - 1] + xs[n // 2])
    return {"median": med, "min": xs[0], "max": xs[-1], "n": n}


_T0 = time.perf_counter()


def _trace(label):
    """BENCH_TRACE=1: where a run's wall time goes, per rank, on stderr (development)."""
    if os.environ.get("BENCH_TRACE"):
        sys.stderr.write("bench.py trace rank %s %7.2f s  %s\n" % (os.environ.get("RANK", "0"), time.perf_counter() - _T0, label))


def main_single_process(args, auto_steps=False):
    """`--gpus N --single-process`: the same job as one process per GPU, driven from ONE process through the library's
    ShardedPlan (pyfft_amd/sharded.py): one plan, stream and scratch per device, the global batch cut into contiguous slices, every
    shard's execute enqueued asynchronously, one synchronisation per device on either side of the K timed steps.  No collective and
    no torch.distributed.  Same JSON line (`n_gpus` = N, `config.parallelism` names the orchestration), so that an 8-GPU box can
    A/B the two orchestrations without a code change.  --share-gpu: shard r on device r % (visible devices)."""
    import numpy
    shape, dtname, batch, seed = CONFIGS[args.config]
    if args.batch:
        batch = args.batch
    dtype = numpy.dtype(dtname)
    split = dtype.kind == "f"
    cdtype = numpy.dtype(numpy.complex64 if dtype in (numpy.complex64, numpy.float32) else numpy.complex128)
    size = int(numpy.prod(shape))
    log2n = sum(int(round(numpy.log2(v))) for v in shape)
    flop_per_xform = 5.0 * size * log2n
    alg_bytes_per_xform = 2.0 * size * cdtype.itemsize
    cpu = None
    if not args.no_cpu_baseline:
        _, _, _, host0 = make_host_block(shape, dtname, batch, seed, 0)
        cpu = cpu_baseline(shape, dtname, host0[:min(len(host0), 16)], flop_per_xform, budget_s=args.cpu_budget, max_workers=args.cpu_workers)
        del host0
    try:
        import torch  # noqa: F401  (first, so that this process uses one HIP runtime for torch and libmifft)
    except Exception:
        pass
    from pyfft_amd import _native as N
    from pyfft_amd.hip import DeviceArray, Event, device_count, device_props
    from pyfft_amd.sharded import ShardedPlan, _OnDevice
    import pyfft_amd.hip as hip
    ndev = device_count()
    if ndev < 1:
        raise SystemExit("bench.py: no GPU visible")
    n = args.gpus
    note = None
    if args.share_gpu:
        devices = [r % ndev for r in range(n)]
    else:
        if n > ndev:
            note = "requested %d GPUs, %d visible: ran %d shards" % (n, ndev, ndev)
            n = ndev
        devices = list(range(n))
    splan = ShardedPlan(shape if len(shape) > 1 else shape[0], dtype, devices=devices, threads=args.shard_threads, wait_for_finish=False)
    gb = batch * n
    slices = splan.slices(gb)
    nplanes = 2 if split else 1
    ins = [[None] * n for _ in range(nplanes)]
    outs = [[None] * n for _ in range(nplanes)]
    blocks = []
    for i, ((start, count), d) in enumerate(zip(slices, devices)):
        blk, host_re, host_im, host_c = make_host_block(shape, dtname, batch, seed, start)
        blocks.append((blk, host_c))
        with _OnDevice(hip, d):
            for pl, host in enumerate((host_re, host_im) if split else (host_c,)):
                ins[pl][i] = DeviceArray((size * count,), dtype)
                outs[pl][i] = ins[pl][i] if args.inplace else DeviceArray((size * count,), dtype)
                fill_device(N, ins[pl][i].ptr, ins[pl][i].nbytes, host)
        del host_re, host_im

    def step():
        if args.inplace:
            splan.execute(*ins, batch=gb, wait_for_finish=False)
        else:
            splan.execute(*(ins + outs), batch=gb, wait_for_finish=False)

    def sync_all():
        for d in sorted(set(devices)):
            with _OnDevice(hip, d):
                N.check(N.lib.mifft_device_sync(), "sync")

    # ---- parity gate: first / middle / last transform of EVERY shard against numpy on the global data set
    parity = None
    rank_report = []
    if not args.inplace:
        step()
        splan.finish()
        eps, mx = (1.1e-6, 1e-5) if cdtype == numpy.complex64 else (1e-11, 1e-10)
        worst_diff = worst_max = 0.0
        isz = dtype.itemsize
        for i, ((start, count), d) in enumerate(zip(slices, devices)):
            blk, host_c = blocks[i]
            sd = 0.0
            with _OnDevice(hip, d):
                for sidx in sorted(set([0, count // 2, count - 1])):
                    parts = []
                    for pl in range(nplanes):
                        got = numpy.empty(size, dtype)
                        N.check(N.lib.mifft_memcpy_d2h(got.ctypes.data, outs[pl][i].ptr + sidx * size * isz, size * isz, None), "d2h")
                        parts.append(got)
                    got = (parts[0].astype(numpy.complex128) + 1j * parts[1]) if split else parts[0].astype(numpy.complex128)
                    ref = numpy.fft.fftn(host_c[sidx % blk].astype(numpy.complex128)).reshape(-1)
                    sd = max(sd, float(numpy.abs(ref - got).sum() / numpy.abs(ref).sum()))
                    worst_max = max(worst_max, float(numpy.abs(ref - got).max() / numpy.abs(ref).max()))
                    if args.dump_dir and sidx in (0, count - 1):
                        numpy.save(os.path.join(args.dump_dir, "xform_%d.npy" % (start + sidx)), got)
            worst_diff = max(worst_diff, sd)
            rank_report.append({"rank": i, "first_transform": start, "count": count, "parity_ok": bool(sd < eps), "difference": sd, "device": d})
        parity = {"samples": 3 * n, "difference": worst_diff, "max_rel": worst_max, "tol_difference": eps, "tol_max_rel": mx,
                  "ok": bool(worst_diff < eps and worst_max <= mx)}
        if not parity["ok"]:
            raise SystemExit("PARITY FAILURE: %r" % (parity,))

    def timed(k):
        """device time of k steps: HIP events on every shard's own stream, the slowest shard counts"""
        ev = []
        for s_ in splan.streams:
            ev.append((Event().record(s_), Event()))
        for _ in range(k):
            step()
        for (e0, e1), s_ in zip(ev, splan.streams):
            e1.record(s_)
        ms = []
        for e0, e1 in ev:
            e1.synchronize()
            ms.append(e1.time_since(e0))
        return max(ms)

    est_step_ms = timed(2) / 2
    if not args.no_spin_up:
        k = max(2, min(400, int(100.0 / max(1e-3, est_step_ms))))
        est_step_ms = timed(k) / k
    if auto_steps:
        args.steps = max(10, int(250.0 / max(1e-3, est_step_ms)) + 1)
    for _ in range(args.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    dev_ms = timed(args.steps)
    sync_all()
    elapsed = time.perf_counter() - t0
    splan.finish()           # raises if a persistent kernel of any shard reported a dependency time-out

    strategies = splan.strategy(gb, inplace=bool(args.inplace))
    total_xforms = gb * args.steps
    ms_per_step = elapsed * 1e3 / args.steps
    gflops = flop_per_xform * total_xforms / elapsed / 1e9
    alg_gbs = alg_bytes_per_xform * total_xforms / elapsed / 1e9
    chain_ms = dev_ms / args.steps
    achieved = alg_bytes_per_xform * batch / (chain_ms * 1e-3) / 1e9          # per device: the slowest shard's kernel
    props = device_props(devices[0])
    result = {
        "metric": "batched_c2c_fft_gflops_5NlogN_1d_n2^20" if args.config == "c2" else "batched_c2c_fft_gflops_5NlogN_" + args.config,
        "value": gflops, "unit": "GFLOPS", "n_gpus": n, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32" if cdtype == numpy.complex64 else "f64",
        "data": "synthetic",
        "config": {"workload": "%s: %s c2c %s, batch %d per GPU, %s, %s" % (
            args.config, "x".join(map(str, shape)), dtname, batch, "split re/im planes" if split else "interleaved",
            "in place" if args.inplace else "out of place"),
            "global_batch": gb, "first_transform_of_rank0": 0,
            "parallelism": "batch-sharded x%d from ONE process (pyfft_amd.sharded.ShardedPlan: a plan and a stream per device, no "
                           "collective, enqueue %s)" % (n, "on a host thread per shard" if args.shard_threads else "on the calling thread"),
            "ranks": rank_report, "passes": [repr(p) for p in splan.plans[0].pass_list(inplace=bool(args.inplace))],
            "strategy": strategies[0][0] if strategies[0] else None, "devices": devices},
        "transforms_per_s": total_xforms / elapsed,
        "algorithmic_GBps": alg_gbs,
        "hbm_fraction_of_8TBps": alg_gbs / n / HBM_PEAK_GBS,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": None, "kernel": "%s on every shard (the slowest shard's HIP-event time)" % (strategies[0][0] if strategies[0] else "-"),
                     "algorithmic_bytes_per_step": alg_bytes_per_xform * batch, "chain_ms_hip_events": chain_ms},
        "protocol": None,
        "parity": parity,
        "device": "%s (%s), %d CUs" % (props.name.decode(), props.gcn_arch.decode(), props.compute_units),
        "cpu_baseline": cpu,
    }
    if note:
        result["note"] = note
    print(json.dumps(result))
    splan.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed steps (default 10; config c5: the 32 chunks of the per-GPU share)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS.keys()))
    ap.add_argument("--batch", type=int, default=0, help="override the per-GPU batch (development only)")
    ap.add_argument("--inplace", action="store_true", help="the K timed steps run in place (the default line times out of place)")
    ap.add_argument("--repeats", type=int, default=5, help="repeats of the out-of-place / in-place protocol blocks (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=24.0, help="seconds of host time the CPU baseline may take (default 24)")
    ap.add_argument("--cpu-workers", type=int, default=0, help="cap on the host cores the CPU baseline uses (default 0 = all of them; tests)")
    ap.add_argument("--plain", action="store_true", help="only parity + warm-up + the K timed steps (no protocol repeats, per-pass timing or CPU baseline): profiler runs")
    ap.add_argument("--chunk-only", action="store_true",
                    help="config c5: time K executes of ONE resident 256-transform chunk (8 GiB in + 8 GiB out, out of place) instead of "
                         "the stated per-GPU share of 8192 transforms (profiler runs, rounds 1-4 lines)")
    ap.add_argument("--selftest-dist", action="store_true", help="CPU/gloo self-test of the multi-process harness")
    ap.add_argument("--force-dist", action="store_true", help="initialise the RCCL process group even at world size 1 (test)")
    ap.add_argument("--control", default="nccl", choices=["nccl", "gloo"],
                    help="backend of the CONTROL plane (barrier, max over ranks); the data path has no collective.  gloo lets "
                         "several ranks share one GPU")
    ap.add_argument("--dump-dir", default=None,
                    help="every rank writes the output of the first and the last transform of its slice there as "
                         "xform_<global index>.npy (test: checked against numpy on the global dataset)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rank r uses device r %% (visible devices): runs the real sharded path with more ranks than GPUs (test)")
    ap.add_argument("--single-process", action="store_true",
                    help="--gpus N driven from ONE process through pyfft_amd.sharded.ShardedPlan (one plan, stream and scratch per "
                         "device, every shard enqueued asynchronously, one sync per device) instead of one process per GPU: the same "
                         "JSON line, `config.parallelism` says which orchestration ran")
    ap.add_argument("--shard-threads", action="store_true", help="--single-process: one host thread per shard for the enqueue")
    ap.add_argument("--no-spin-up", action="store_true", help="skip the untimed clock spin-up in front of the warm-up steps (A/B)")
    args = ap.parse_args()

    if args.plain:
        args.repeats = 0
        args.no_cpu_baseline = True
    # config 5 as BASELINE.json states it: 65536 transforms of 2^22 points over 8 GPUs = 8192 per GPU = 256 GiB per GPU, run as a
    # streaming loop over 256-transform chunks.  One step = one chunk; the default K is one sweep over the share.
    share_mode = args.config == "c5" and not args.chunk_only and not args.batch and not args.single_process
    auto_steps = args.steps <= 0 and not share_mode      # K chosen after the warm-up so that the timed region is >= 250 ms
    if args.steps <= 0:
        args.steps = C5_SHARE // CONFIGS["c5"][2] if share_mode else 10
    if args.single_process:
        return main_single_process(args, auto_steps)
    if args.gpus > 1 and "RANK" not in os.environ:
        return self_launch(args, sys.argv[1:])       # never returns
    if args.selftest_dist:
        return selftest_dist(args)

    rank, local_rank, world = dist_env()
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))

    import numpy
    shape, dtname, batch, seed = CONFIGS[args.config]
    if args.batch:
        batch = args.batch
    dtype = numpy.dtype(dtname)
    split = dtype.kind == "f"
    cdtype = numpy.dtype(numpy.complex64 if dtype in (numpy.complex64, numpy.float32) else numpy.complex128)
    size = int(numpy.prod(shape))
    log2n = sum(int(round(numpy.log2(s))) for s in shape)
    flop_per_xform = 5.0 * size * log2n
    alg_bytes_per_xform = 2.0 * size * cdtype.itemsize          # read once + write once
    share = C5_SHARE if share_mode else batch                    # transforms this rank owns (share mode: `batch` is the chunk)
    gstart, _ = shard_batch(share * world, rank, world)          # this rank's slice of the global batch

    # ---- synthetic data: one host block of <= 64 transforms (tiled across the batch on the device further down)
    blk, host_re, host_im, host_c = make_host_block(shape, dtname, batch, seed, gstart)
    _trace("host block made")

    # ---- CPU baseline first: rank 0 (at every world size: the other ranks wait in init_process_group), before this process
    # initialises the GPU (the pool forks) -- numpy.fft on the box's host cores in the same run, as north_star words it
    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        cpu = cpu_baseline(shape, dtname, host_c[:min(blk, 16)], flop_per_xform, budget_s=args.cpu_budget, max_workers=args.cpu_workers)
    _trace("cpu baseline done (rank 0 only)")

    torch = None
    try:
        import torch  # first, so that this process uses one HIP runtime for torch and libmifft
    except Exception:
        if world > 1:
            raise
    dist = None
    device = local_rank
    if args.share_gpu:
        from pyfft_amd.hip import device_count
        device = local_rank % max(1, device_count())
    if torch is not None and torch.cuda.is_available():
        torch.cuda.set_device(device)
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:            # --force-dist at world 1 without a launcher
            os.environ["MASTER_ADDR"] = "127.0.0.1"
            os.environ["MASTER_PORT"] = str(_free_port())
        dist.init_process_group(args.control, rank=rank, world_size=world)
    _trace("process group up")

    from pyfft_amd import _native as N
    from pyfft_amd.hip import Plan, DeviceArray, Event, device_props
    from pyfft_amd.passes import launch_units
    N.check(N.lib.mifft_set_device(device), "set_device")
    props = device_props(device)

    nel = size * batch
    resident = False
    chunks = 1
    if share_mode:
        # the whole per-GPU share resident, transformed IN PLACE chunk by chunk (256 GiB of the 288 GB; out of place would need 512)
        chunks = share // batch
        try:
            ins = [DeviceArray((size * share,), dtype)]
            outs = ins
            resident = True
            fill_device(N, ins[0].ptr, ins[0].nbytes, host_c)
        except RuntimeError as e:
            # no room (a shared or smaller device): the streaming loop over ONE pair of reused chunk buffers, out of place
            sys.stderr.write("bench.py: the %d-transform share does not fit (%s): reusing one resident chunk\n" % (share, str(e)[:120]))
            ins = None
    if share_mode and resident:
        pass
    elif split:
        ins = [DeviceArray((nel,), dtype), DeviceArray((nel,), dtype)]
        outs = ins if args.inplace else [DeviceArray((nel,), dtype), DeviceArray((nel,), dtype)]
        fill_device(N, ins[0].ptr, ins[0].nbytes, host_re)
        fill_device(N, ins[1].ptr, ins[1].nbytes, host_im)
    else:
        ins = [DeviceArray((nel,), dtype)]
        outs = ins if args.inplace else [DeviceArray((nel,), dtype)]
        fill_device(N, ins[0].ptr, ins[0].nbytes, host_c)
    del host_re, host_im

    plan = Plan(shape if len(shape) > 1 else shape[0], dtype=dtype, wait_for_finish=False)
    # every execute() of this process is counted (parity gate, spin-up, warm-up, timed steps ...): the profiler runs of
    # tools/pmc_traffic.py divide the counters of the whole process by this number
    executes_total = [0]
    _plan_execute = plan.execute

    def _counted_execute(*a, **k):
        executes_total[0] += 1
        return _plan_execute(*a, **k)
    plan.execute = _counted_execute
    _trace("device buffers filled, plan built")
    stream = plan._context.getQueue()

    def execute(inplace, inverse=False):
        if split:
            if inplace:
                plan.execute(ins[0], ins[1], batch=batch, inverse=inverse)
            else:
                plan.execute(ins[0], ins[1], outs[0], outs[1], batch=batch, inverse=inverse)
        else:
            if inplace:
                plan.execute(ins[0], batch=batch, inverse=inverse)
            else:
                plan.execute(ins[0], outs[0], batch=batch, inverse=inverse)

    chunk_bytes = size * batch * cdtype.itemsize

    def execute_chunk(c, inverse=False):
        """share mode, resident: chunk c of the share (global transforms [gstart + c * batch, gstart + (c + 1) * batch)) in place"""
        plan.execute(ins[0].ptr + c * chunk_bytes, batch=batch, inverse=inverse)

    step_no = [0]

    def step():
        if resident:
            # one sweep over the share forward, the next one inverse, ...: the values stay those of the data set
            i = step_no[0]
            step_no[0] = i + 1
            execute_chunk(i % chunks, inverse=bool((i // chunks) & 1))
        else:
            execute(args.inplace)

    def sync_all():
        if torch is not None and torch.cuda.is_available():
            torch.cuda.synchronize()
        else:
            N.check(N.lib.mifft_device_sync(), "sync")

    def barrier():
        if dist is not None:
            dist.barrier()

    # ---- parity gate (before timing, on the untouched input): sampled batch items vs numpy.fft
    parity = None
    if resident:
        # every chunk of the share forward in place; first / middle / last transform of the first, a middle and the last chunk against
        # numpy; then every chunk inverse in place (the data set is back, up to rounding) and the round trip of the same samples
        isz = dtype.itemsize
        eps, mx = 1.1e-6, 1e-5

        def fetch(g):
            got = numpy.empty(size, dtype)
            N.check(N.lib.mifft_memcpy_d2h(got.ctypes.data, ins[0].ptr + g * size * isz, size * isz, None), "d2h")
            return got.astype(numpy.complex128)

        samples = sorted(set(c * batch + s for c in (0, chunks // 2, chunks - 1) for s in (0, batch // 2, batch - 1)))
        for c in range(chunks):
            execute_chunk(c)
        plan.finish()
        worst_diff, worst_max = 0.0, 0.0
        for g in samples:
            got = fetch(g)
            ref = numpy.fft.fft(host_c[g % blk].astype(numpy.complex128))
            worst_diff = max(worst_diff, float(numpy.abs(ref - got).sum() / numpy.abs(ref).sum()))
            worst_max = max(worst_max, float(numpy.abs(ref - got).max() / numpy.abs(ref).max()))
            if args.dump_dir and g in (0, share - 1):
                numpy.save(os.path.join(args.dump_dir, "xform_%d.npy" % (gstart + g)), got)
        for c in range(chunks):
            execute_chunk(c, inverse=True)
        plan.finish()
        worst_rt = 0.0
        for g in samples:
            want = host_c[g % blk].astype(numpy.complex128)
            worst_rt = max(worst_rt, float(numpy.abs(want - fetch(g)).sum() / numpy.abs(want).sum()))
        parity = {"samples": len(samples), "sampled_transforms": samples, "difference": worst_diff, "max_rel": worst_max,
                  "round_trip_difference": worst_rt, "tol_difference": eps, "tol_max_rel": mx,
                  "ok": bool(worst_diff < eps and worst_max <= mx and worst_rt < eps)}
        if not parity["ok"]:
            raise SystemExit("PARITY FAILURE: %r" % (parity,))
    elif not args.inplace:
        step()
        plan.finish()
        # >= 8 sampled batch items including the first and the last one (all of them when the batch is smaller)
        samples = sorted(set([0, 1 % batch, 2 % batch, (blk - 1) % batch, blk % batch, batch // 3, batch // 2, (2 * batch) // 3,
                              (batch - blk) % batch, max(0, batch - 2), batch - 1]))
        worst_diff, worst_max = 0.0, 0.0
        isz = dtype.itemsize
        for s in samples:
            if split:
                re = numpy.empty(size, dtype)
                im = numpy.empty(size, dtype)
                N.check(N.lib.mifft_memcpy_d2h(re.ctypes.data, outs[0].ptr + s * size * isz, size * isz, None), "d2h")
                N.check(N.lib.mifft_memcpy_d2h(im.ctypes.data, outs[1].ptr + s * size * isz, size * isz, None), "d2h")
                got = re.astype(numpy.complex128) + 1j * im
            else:
                got = numpy.empty(size, dtype)
                N.check(N.lib.mifft_memcpy_d2h(got.ctypes.data, outs[0].ptr + s * size * isz, size * isz, None), "d2h")
                got = got.astype(numpy.complex128)
            ref = numpy.fft.fftn(host_c[s % blk].astype(numpy.complex128)).reshape(-1)
            worst_diff = max(worst_diff, float(numpy.abs(ref - got).sum() / numpy.abs(ref).sum()))
            worst_max = max(worst_max, float(numpy.abs(ref - got).max() / numpy.abs(ref).max()))
            if args.dump_dir and s in (0, batch - 1):
                numpy.save(os.path.join(args.dump_dir, "xform_%d.npy" % (gstart + s)), got)
        eps = 1.1e-6 if cdtype == numpy.complex64 else 1e-11
        mx = 1e-5 if cdtype == numpy.complex64 else 1e-10
        parity = {"samples": len(samples), "difference": worst_diff, "max_rel": worst_max,
                  "tol_difference": eps, "tol_max_rel": mx, "ok": bool(worst_diff < eps and worst_max <= mx)}
        if not parity["ok"]:
            raise SystemExit("PARITY FAILURE: %r" % (parity,))

    _trace("parity gate passed")
    # ---- clock spin-up (untimed, not a step of the contract): a device that has idled through the host-side parity check starts its
    # first kernels at a low clock, and a timed region of a few short steps right behind two warm-up steps reads the ramp (C3: 0.4255 on
    # the K-step line against 0.445 on >= 20 ms blocks).  ~100 ms of back-to-back steps first; their duration also
    # gives the step time the default K and the protocol's block length are sized from.
    def timed_steps(n):
        e0, e1 = Event(), Event()
        e0.record(stream)
        for _ in range(n):
            if resident:
                execute_chunk(0)
                execute_chunk(0, inverse=True)
            else:
                step()
        e1.record(stream)
        e1.synchronize()
        return e1.time_since(e0) / (n * (2 if resident else 1))

    est_step_ms = timed_steps(2)
    if not args.no_spin_up:
        est_step_ms = timed_steps(max(2, min(400, int(100.0 / max(1e-3, est_step_ms)))))
    if auto_steps:
        args.steps = max(10, int(250.0 / max(1e-3, est_step_ms)) + 1)
    _trace("spin-up done: %.3f ms per step, K = %d" % (est_step_ms, args.steps))
    # ---- the reference's timing protocol (untimed for `value`; run BEFORE the timed steps since round 6: it is also what keeps the
    # clocks up until they start): out of place AND in place, median of >= 5 repeats of a
    # block of back-to-back executes between two HIP events (test/test_performance.py:22-30 times 10 executes once;
    # cuda/test.cu:37-64 times both forms).  In place alternates forward / inverse so that the values stay bounded.
    protocol = None
    if resident and args.repeats > 0:
        # the same share as ONE in-place execute (batch = 8192: element offsets beyond 2^35), forward then inverse, HIP events
        protocol = {"one_execute_of_the_share": {"batch": share, "repeats": min(args.repeats, 3)}}
        ms = []
        for _ in range(min(args.repeats, 3)):
            for inv in (False, True):
                e0, e1 = Event(), Event()
                e0.record(stream)
                plan.execute(ins[0], batch=share, inverse=inv)
                e1.record(stream)
                e1.synchronize()
                ms.append(e1.time_since(e0))
        plan.finish()
        st = stats(ms)
        protocol["one_execute_of_the_share"].update({
            "ms_per_execute": st, "strategy": plan.strategy(share)[0],
            "frac_median": alg_bytes_per_xform * share / (st["median"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "transforms_per_s_median": share / (st["median"] * 1e-3)})
    elif args.repeats > 0 and world == 1 and not args.inplace:
        # blocks of >= 20 ms of back-to-back executes (a burst of a few short launches behind a synchronisation measures the clock
        # ramp and the fill / drain of the first launch: 3-4 points at 1 GiB per side, docs/measurement.md)
        per = max(2, min(args.steps, 10), int(20.0 / max(1e-3, est_step_ms)) + 1)
        per += per & 1
        protocol = {"executes_per_repeat": per, "repeats": args.repeats}
        for name, inpl in (("out_of_place", False), ("in_place", True)):
            ms = []
            for _ in range(args.repeats):
                e0, e1 = Event(), Event()
                e0.record(stream)
                for j in range(per):
                    execute(inpl, inverse=bool(inpl and (j & 1)))
                e1.record(stream)
                e1.synchronize()
                ms.append(e1.time_since(e0) / per)
            plan.finish()
            st = stats(ms)
            protocol[name] = {"ms_per_execute": st,
                              "gflops_median": flop_per_xform * batch / (st["median"] * 1e-3) / 1e9,
                              "frac_median": alg_bytes_per_xform * batch / (st["median"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "frac_min_max": [alg_bytes_per_xform * batch / (st["max"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                               alg_bytes_per_xform * batch / (st["min"] * 1e-3) / 1e9 / HBM_PEAK_GBS],
                              "strategy": plan.strategy(batch, inplace=inpl)[0]}

    # ---- warm-up, then EXACTLY K timed steps bracketed by barrier + device sync on both sides
    for w in range(args.warmup):
        if resident:
            execute_chunk(w % chunks)
            execute_chunk(w % chunks, inverse=True)
        else:
            step()
    sync_all()
    barrier()
    sync_all()
    ev0, ev1 = Event(), Event()
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        step()
    ev1.record(stream)
    sync_all()
    barrier()
    sync_all()
    elapsed = time.perf_counter() - t0
    dev_ms = ev1.time_since(ev0)          # HIP events on the stream the kernels are launched on
    plan.finish()                          # raises if a persistent kernel reported a dependency time-out (results invalid)

    rank_report = None
    if dist is not None:
        ctl_dev = "cuda" if args.control == "nccl" else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=ctl_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # every rank's slice of the global batch and its own parity verdict (checked against the GLOBAL dataset)
        mine = torch.tensor([float(gstart), float(share), 1.0 if (parity is None or parity["ok"]) else 0.0,
                             parity["difference"] if parity else 0.0, float(device)], dtype=torch.float64, device=ctl_dev)
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        rank_report = [{"rank": r, "first_transform": int(g[0].item()), "count": int(g[1].item()), "parity_ok": bool(g[2].item() > 0.5),
                        "difference": float(g[3].item()), "device": int(g[4].item())} for r, g in enumerate(gathered)]

    _trace("timed steps done, ranks gathered")
    # ---- per-pass device time (separate, untimed-for-value measurement; HIP events between launches).
    # Only meaningful for the one-launch-per-pass strategy; the fused / pipelined strategies own a small scratch.
    import ctypes
    timed_inplace = bool(args.inplace or resident)
    strategy = plan.strategy(batch, inplace=timed_inplace)     # (out of place: ("nd_oop",) where the one-launch kernel bypasses the chain)
    timed_passes = plan.pass_list(inplace=timed_inplace)
    npass = len(timed_passes)
    pass_ms = None
    if strategy[0] == "chain" and not args.plain:
        descs = plan._descriptors(batch, args.inplace, False)
        ptr = plan._context.pointer_of
        if split:
            b0 = [ins[0].ptr, outs[0].ptr, ptr(plan._tempmemobj) if plan._tempmemobj is not None else None]
            b1 = [ins[1].ptr, outs[1].ptr, None]
        else:
            b0 = [ins[0].ptr, outs[0].ptr, ptr(plan._tempmemobj) if plan._tempmemobj is not None else None]
            b1 = [None, None, None]
        pass_ms = []
        reps = 3
        # one entry per LAUNCH of the plan: a pass, or a pass pair (two descriptors, one kernel; both carry the unit's src / dst)
        i = 0
        for unit, count in launch_units(plan.pass_list()):
            d = descs[i]
            dl = descs[i + count - 1]
            e0, e1 = Event(), Event()
            e0.record(stream)
            for _ in range(reps):
                if count == 2:
                    N.check(N.lib.mifft_launch_pass_pair(ctypes.byref(d), ctypes.byref(dl), b0[d.src], b1[d.src], b0[dl.dst], b1[dl.dst],
                                                         plan._context.stream_handle()), "launch_pass_pair")
                else:
                    N.check(N.lib.mifft_launch_pass(ctypes.byref(d), b0[d.src], b1[d.src], b0[d.dst], b1[d.dst],
                                                    plan._context.stream_handle()), "launch_pass")
            e1.record(stream)
            e1.synchronize()
            pass_ms.append(e1.time_since(e0) / reps)
            i += count
    nlaunch = len(launch_units(timed_passes))
    if strategy[0] in ("chain", "nd_oop"):
        launches = "%d launches per step" % nlaunch
    elif strategy[0] in ("fused2", "fused2x", "fusedp"):
        what = {"fused2": "both passes", "fused2x": "both passes, one work list per XCD", "fusedp": "both pass pairs"}[strategy[0]]
        launches = "1 persistent launch per step (%s, lag %s, ring %s)" % ((what,) + tuple(strategy[1:3]))
    elif strategy[0] == "fused2z":
        launches = "1 persistent launch over the (y, x) planes (lag %s, ring %s) + %d plain z launch(es) per step" % (tuple(strategy[1:3]) + (nlaunch - 2,))
    elif strategy[0] == "xcd2":
        launches = "1 persistent launch per step (both passes, XCD-resident intermediate)"
    else:
        launches = "%d launches per chunk, chunks of %s items on %s streams" % ((nlaunch,) + tuple(strategy[1:3]))

    total_xforms = batch * world * args.steps
    ms_per_step = elapsed * 1e3 / args.steps
    gflops = flop_per_xform * total_xforms / elapsed / 1e9
    alg_gbs = alg_bytes_per_xform * total_xforms / elapsed / 1e9
    chain_ms = dev_ms / args.steps
    achieved = alg_bytes_per_xform * batch / (chain_ms * 1e-3) / 1e9

    # HBM traffic of one step from the PMC counters: profiles/traffic_<config>.json, regenerated by tools/pmc_traffic.py
    # (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this very command; the file names its source run)
    # (config 5: the file is measured on the --chunk-only form, the same kernel on the same 256-transform chunk)
    traffic = None
    traffic_source = None
    tf = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.config)
    if os.path.exists(tf) and not args.batch and not args.inplace:   # measured on the default workload only
        try:
            tj = json.load(open(tf))
            if tj.get("strategy", strategy[0]) == strategy[0]:
                traffic = tj.get("hbm_bytes_per_step")
                lib_now = library_digest()
                traffic_source = {"file": "profiles/traffic_%s.json" % args.config, "commit": tj.get("commit"),
                                  "libmifft_sha256_16": tj.get("libmifft_sha256_16"),
                                  "measured_on_the_running_library": (tj.get("libmifft_sha256_16") == lib_now) if tj.get("libmifft_sha256_16") else None,
                                  "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --config %s --plain` "
                                            "(tools/pmc_traffic.py), not collected in this run" % args.config}
        except Exception:
            traffic = None

    c5_note = None
    if args.config == "c5" and not args.batch:
        # BASELINE.json configs[4]: 65536 transforms of 2^22 points over 8 GPUs = 8192 per GPU = 256 GiB per GPU, run as a streaming
        # loop over 256-transform chunks: one step of this line is ONE chunk
        c5_note = {"chunk_transforms": batch, "chunks_per_gpu_for_stated_config": C5_SHARE // batch,
                   "ms_per_gpu_for_stated_config": ms_per_step * (C5_SHARE // batch),
                   "transforms_per_s_per_gpu": batch / (ms_per_step * 1e-3),
                   "stated_config": "1-D c2c fp32 N=2^22 batch=65536 sharded across 8 GPUs (8192 per GPU)",
                   "form": ("the rank's whole share of %d transforms (256 GiB) resident in HBM, transformed in place chunk by chunk: step i = "
                            "chunk i %% %d, sweeps alternate forward / inverse" % (share, chunks)) if resident else
                           ("ONE resident chunk, out of place, executed K times (--chunk-only)" if not share_mode else
                            "the share did not fit this device: ONE pair of reused chunk buffers, out of place, executed K times"),
                   "share_resident": bool(resident)}
    result = {
        "metric": "batched_c2c_fft_gflops_5NlogN_1d_n2^20" if args.config == "c2" else "batched_c2c_fft_gflops_5NlogN_" + args.config,
        "value": gflops,
        "unit": "GFLOPS",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32" if cdtype == numpy.complex64 else "f64",
        "data": "synthetic",
        "config": {"workload": "%s: %s c2c %s, batch %s per GPU, %s, %s" % (
            args.config, "x".join(map(str, shape)), dtname, ("%d in chunks of %d" % (share, batch)) if share_mode else str(batch),
            "split re/im planes" if split else "interleaved", "in place" if (args.inplace or resident) else "out of place"),
            "global_batch": share * world, "first_transform_of_rank0": gstart,
            "parallelism": "batch-sharded x%d, no collective%s" % (world, " (control plane: %s)" % args.control if dist is not None else ""),
            "ranks": rank_report,
            "passes": [repr(p) for p in timed_passes], "strategy": strategy[0]},
        "transforms_per_s": total_xforms / elapsed,
        "executes_in_this_process": executes_total[0],
        "algorithmic_GBps": alg_gbs,
        "hbm_fraction_of_8TBps": alg_gbs / world / HBM_PEAK_GBS,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS,
                     # (the protocol's median over >= 20 ms blocks of the same executes, for lines whose K steps are short)
                     "frac_protocol_median": (protocol or {}).get("in_place" if timed_inplace else "out_of_place", {}).get("frac_median"),
                     "traffic": traffic, "traffic_source": traffic_source,
                     "kernel": "%s: %s" % (strategy[0], launches),
                     "algorithmic_bytes_per_step": alg_bytes_per_xform * batch,
                     "chain_ms_hip_events": chain_ms, "pass_ms_hip_events": pass_ms},
        "protocol": protocol,
        "parity": parity,
        "device": "%s (%s), %d CUs" % (props.name.decode(), props.gcn_arch.decode(), props.compute_units),
        "cpu_baseline": cpu,
    }
    if c5_note is not None:
        result["config"].update(c5_note)
    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

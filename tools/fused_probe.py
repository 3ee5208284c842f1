"""Probe of the fused two-pass persistent kernel (mifft_launch_fused2): correctness against the two-launch chain
(bit-identical expected) and a sweep of lag / ring / grid."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
from pyfft_amd.hip import Plan, DeviceArray, Event, DeviceAllocation
from pyfft_amd import _native as N

def fill(b):
    rng = numpy.random.default_rng(1)
    host = rng.standard_normal(1 << 24).astype(numpy.float32)
    N.check(N.lib.mifft_memcpy_h2d(b.ptr, host.ctypes.data, min(host.nbytes, b.nbytes), None))
    done = min(host.nbytes, b.nbytes)
    while done < b.nbytes:
        n = min(done, b.nbytes - done)
        N.check(N.lib.mifft_memcpy_d2d(b.ptr + done, b.ptr, n, None))
        done += n
    N.check(N.lib.mifft_device_sync())

def probe(n, B, combos, iters=3, check=True):
    dtype = numpy.complex64; isz = 8
    nel = n * B
    a = DeviceArray((nel,), dtype); b = DeviceArray((nel,), dtype); ref = DeviceArray((nel,), dtype)
    fill(a)
    plan = Plan((n,), dtype=dtype, wait_for_finish=False)
    st = plan._context.getQueue()
    plan.execute(a, ref, batch=B); st.synchronize()
    best = 1e9
    for _ in range(iters):
        e0 = Event().record(st); plan.execute(a, ref, batch=B); e1 = Event().record(st); e1.synchronize()
        best = min(best, e1.time_since(e0))
    alg = 2.0 * nel * isz
    print("N=%d B=%d two launches %s: %.3f ms  %.1f%% of 8 TB/s" % (n, B, plan.pass_list(), best, alg / best / 1e6 / 80), flush=True)
    descs = plan._descriptors(B, False, False)
    refh = ref.get() if check else None
    counters = DeviceAllocation(N.fused2_counter_bytes(B))
    sync = N.MifftFusedSync(counters.ptr, None, None)       # one counter set, zeroed by the call; error word = counters[1]
    for lag, ring, grid in combos:
        scratch = DeviceAllocation(ring * n * isz)
        def once():
            N.check(N.lib.mifft_launch_fused2(ctypes.byref(descs[0]), ctypes.byref(descs[1]), a.ptr, None, b.ptr, None,
                                              scratch.ptr, None, ring, lag, ctypes.byref(sync), grid, st.handle), "fused2")
        N.check(N.lib.mifft_memset(b.ptr, 0, b.nbytes, st.handle))
        once(); st.synchronize()
        cnt = numpy.zeros(2, numpy.uint32)
        N.check(N.lib.mifft_memcpy_d2h(cnt.ctypes.data, counters.ptr, 8, None))
        ok = "unchecked"
        if check:
            ok = "bit-identical" if numpy.array_equal(b.get(), refh) else "MISMATCH"
        best = 1e9
        for _ in range(iters):
            e0 = Event().record(st); once(); e1 = Event().record(st); e1.synchronize()
            best = min(best, e1.time_since(e0))
        print("  lag=%-2d ring=%-3d grid=%-4d (scratch %4.0f MiB): %.3f ms  %.1f%% of 8 TB/s  %s  err=%d" % (
            lag, ring, grid, ring * n * isz / 2**20, best, alg / best / 1e6 / 80, ok, cnt[1]), flush=True)
        del scratch

if __name__ == "__main__":
    combos = [(7, 14, 512), (8, 16, 512), (9, 18, 512), (10, 20, 512), (8, 12, 512), (8, 24, 512), (8, 16, 384), (8, 16, 448), (6, 16, 384)]
    if len(sys.argv) > 1 and sys.argv[1] == "r2":     # end of round 2 (counters on their own lines): around the plan's lag 14 / ring 28
        combos = [(14, 28, 512), (10, 28, 512), (18, 28, 512), (7, 28, 512), (12, 24, 512), (10, 20, 512), (7, 14, 512), (4, 8, 512), (16, 32, 512), (14, 28, 768)]
        probe(1 << 20, 2048, combos, check=False)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "wide":
        combos = [(l, r, 512) for l in (6, 8, 10, 12, 14, 18, 24) for r in (2 * l, 3 * l // 2)] + [(18, 36, 768), (12, 24, 768), (18, 36, 384)]
    probe(1 << 20, 1024, combos)

"""Probe of the stream-pipelined chunked execution (mifft_launch_chain_pipelined): sweep chunk size and
number of side streams for N = 2^20 c64 and check the result against the plain chain."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy
from pyfft_amd.hip import Plan, DeviceArray, Event, Stream, DeviceAllocation
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

def probe(shape, B, combos, dtype=numpy.complex64, iters=3):
    size = int(numpy.prod(shape)); nel = size * B
    isz = numpy.dtype(dtype).itemsize
    a = DeviceArray((nel,), dtype); b = DeviceArray((nel,), dtype); ref = DeviceArray((nel,), dtype)
    fill(a)
    plan = Plan(shape, dtype=dtype, wait_for_finish=False)
    st = plan._context.getQueue()
    plan.execute(a, ref, batch=B); st.synchronize()
    best = 1e9
    for _ in range(iters):
        e0 = Event().record(st); plan.execute(a, ref, batch=B); e1 = Event().record(st); e1.synchronize()
        best = min(best, e1.time_since(e0))
    alg = 2.0 * nel * isz
    print("%s B=%d plain chain: %.3f ms  %.1f%% of 8 TB/s" % (shape, B, best, alg / best / 1e6 / 80), flush=True)
    descs = plan._descriptors(B, False, False)
    npass = len(plan.pass_list())
    refh = ref.get()
    for chunk, nside in combos:
        side = [Stream() for _ in range(nside)]
        evs = [Event() for _ in range(nside + 1)]
        tmp = DeviceAllocation(nside * chunk * size * isz)
        side_arr = (ctypes.c_void_p * nside)(*[s.handle for s in side])
        ev_arr = (ctypes.c_void_p * (nside + 1))(*[e.handle for e in evs])
        bufs0 = N.make_buf3(a.ptr, b.ptr, tmp.ptr)
        def once():
            N.check(N.lib.mifft_launch_chain_pipelined(descs, npass, bufs0, None, B, chunk, size, st.handle, side_arr, nside, ev_arr), "pipelined")
        N.check(N.lib.mifft_memset(b.ptr, 0, b.nbytes, st.handle))
        once(); st.synchronize()
        ok = numpy.array_equal(b.get(), refh)
        best = 1e9
        for _ in range(iters):
            e0 = Event().record(st); once(); e1 = Event().record(st); e1.synchronize()
            best = min(best, e1.time_since(e0))
        print("  chunk=%-4d streams=%d (scratch %4.0f MiB): %.3f ms  %.1f%% of 8 TB/s  %s" % (
            chunk, nside, nside * chunk * size * isz / 2**20, best, alg / best / 1e6 / 80, "bit-identical" if ok else "MISMATCH"), flush=True)
        del tmp

if __name__ == "__main__":
    combos = [(c, s) for s in (2, 3, 4) for c in (2, 4, 8, 16, 32)]
    probe((1 << 20,), 1024, combos)
    probe((1 << 16,), 16384, [(64, 2), (128, 2), (256, 2), (128, 4), (512, 2)])

// Fused two-pass kernels for a long contiguous axis N = L0 * L1 -- fp32: L0, L1 in {256, 512, 1024} on the 256-thread tiles
// of fft_col2.hpp, 2048 x 2048 / 2048 x 1024 on the 512-thread tiles of fft_col3.hpp; fp64: 1024 x 1024 -- and for the 2-D
// squares 512 / 1024 / 2048 (fp32) and 1024 (fp64): both Stockham
// passes of every transform run inside ONE persistent launch, in dependency order, so the inter-pass
// intermediate of a transform is consumed a few transforms later while it is still in the 256 MiB Infinity
// Cache, and there are no launch boundaries (with 128 KiB tiles a cache-sized chunk is only ~2 machine-waves
// of tiles, so per-chunk launches lose the cache gain to their tails).
//
// Work list (one global ticket counter): group g holds the pass-0 tiles of transform g interleaved with the
// pass-1 tiles of transform g - lag.  Dependencies, all on LOWER ticket numbers (=> no deadlock for any
// dispatch order / placement / residency):
//     pass-1 tile of t   waits until all pass-0 tiles of t have published            (wdone[t] == tiles0)
//     pass-0 tile of t   waits until all pass-1 tiles of t - ring have finished reading the ring slot it
//                        is about to overwrite                                       (rdone[t-ring] == tiles1)
// Hand-off (cdna_hip_programming.md Guideline 16, write-through form): the intermediate is written with
// agent-coherent write-through stores, every storing wave drains vmcnt, the work-group barrier, one lane bumps
// the counter (relaxed, agent scope); the consumer's one lane polls relaxed, ONE acquire fence, vmcnt drain,
// barrier, then plain loads.  (A release fence per tile -- buffer_wbl2 -- made this kernel 2x slower than two
// launches.)  The scratch ring is always interleaved, also for split-plane user buffers (SPLIT: the input of
// pass 0 and the output of pass 1 are two scalar planes).
// Every spin is bounded; on timeout an error word is set (the host checks it) instead of hanging the GPU.
#pragma once
#include "fft_col2.hpp"
#include "fft_col3.hpp"

namespace mifft {

constexpr unsigned kFusedCS = 64u;   // words between two counters (= MIFFT_FUSED2_COUNTER_STRIDE)

struct FusedArgs {
    TileArgs p0;         // pass 0: in = user input,  out = scratch ring (matrix index = ring slot)
    TileArgs p1;         // pass 1: in = scratch ring, out = user output
    unsigned* counters;  // [0] ticket, [1] error, then one counter per CS = 64 words (256 bytes): wdone[t] at CS * (1 + t),
                         // rdone[t] at CS * (1 + batch + t)  (zeroed per launch)
    unsigned batch;      // number of transforms
    unsigned lag;        // pass 1 of transform t is queued with pass 0 of transform t + lag
    unsigned ring;       // scratch ring slots (transforms); ring > lag
    unsigned tiles0;     // 16-column tiles per transform in pass 0 (= L1 / 16)
    unsigned tiles1;     // 16-column tiles per transform in pass 1 (= L0 / 16)
};

// XCD-local form (development strategy `fusedx`, round 3): one work list PER XCD (chiplet).  A work-group reads its XCD from
// HW_REG_XCC_ID and draws tickets from that XCD's counter; XCD x owns the transforms t = x + 8 i, its ring slots are
// [x * ring, (x + 1) * ring), so a transform's intermediate is written and read by work-groups of ONE XCD and -- written with
// plain stores -- can stay in that XCD's 4 MiB L2 (tools/l2_resident_probe.hip: ~1 MiB stays next to non-temporal streams, 2 MiB
// leaks 44 % of its writes).  Same dependency order per list, so the same no-deadlock argument; an XCD that receives no
// work-group at all would leave its transforms undone, which the launch detects (a census word per XCD, error bit 2).
__device__ __forceinline__ unsigned fused_xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7u;
}

// Deferred publish of a pass-0 tile: its write-through stores drain under the first poll of the NEXT item's dependency, and
// the counter is bumped BEFORE this work-group starts to wait.  (Publishing only after the wait has ended deadlocks: work-group
// X owes a tile of transform T and waits for U while Y owes a tile of U and waits for T -- measured as dependency time-outs.)
struct FusedPending {
    unsigned* ctr;   // wdone counter still to be bumped, or nullptr
};

// wait until *ctr >= target (one lane polls, bounded); ACQ: also make other work-groups' published data visible.
// `seen` is lane 0's EARLY poll of the same counter, issued from the middle of the previous tile (FusedHook below): an agent-scope load takes 1-3 us
// under load, and with the poll on the critical path of every item the kernel lost 16 % (C2: 22.5 ms; with no waits at all --
// wrong results, same traffic -- 18.9 ms).  The counters only grow, so a stale value can only under-estimate.
template <bool ACQ> __device__ __forceinline__ void fused_wait_ge(unsigned* ctr, unsigned target, unsigned seen, unsigned* err, FusedPending& pend) {
    if (pend.ctr != nullptr) {   // (uniform) the owed tile: every wave drains its stores, the barrier joins them, lane 0 publishes
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(pend.ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pend.ctr = nullptr;
    }
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        while (seen < target) {          // (the early value was not enough: look again, then sleep between polls)
            seen = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (seen >= target) break;
            __builtin_amdgcn_s_sleep(32);
            if (++spins > (1u << 22)) {  // ~ seconds: never hang the GPU
                __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        if constexpr (ACQ) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
}

// publish now: every wave drains its own memory operations, the work-group barrier joins them, one lane bumps the counter.
// (The data a later tile depends on was written with write-through stores, so no release fence is needed.)
__device__ __forceinline__ void fused_flush(FusedPending& pend) {
    if (pend.ctr != nullptr) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(pend.ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pend.ctr = nullptr;
    }
}

// "this tile has finished READING" (pass 1 and its ring slot): every load has been consumed by the time the tile's last
// butterflies ran, so the stores of the result are not waited for.
__device__ __forceinline__ void fused_signal_read(unsigned* ctr) {
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Lane 0's view of the work list: the ticket of the next item (drawn at the top of this one) and the early poll of that item's
// dependency counter (issued from the middle of this tile, TileHook below).  Both returning atomics are consumed at the top of
// the next item, so neither latency is on the critical path.
struct FusedQueue {
    unsigned t1, seen1;
};

struct FusedItem {
    unsigned pass;     // 0, 1, or 2 = nothing to do (fill / drain of the pipeline)
    unsigned t;        // transform
    unsigned slot;     // its ring slot
    unsigned tile;
    unsigned* dep;     // counter this item waits for (nullptr: none)
    unsigned target;
};

// xs = 8 and x = the XCD for the XCD-local lists (group g of the list = transform x + 8 g, `nb` = transforms of this list);
// xs = 1, x = 0, nb = batch for the one global list.  it.t is the GLOBAL transform, it.slot its ring slot.
template <unsigned PER0, unsigned PER1>
__device__ __forceinline__ FusedItem fused_decode(const FusedArgs& f, unsigned item, unsigned gsize, unsigned* wdone, unsigned* rdone,
                                                  unsigned x = 0u, unsigned xs = 1u, unsigned nb = 0xffffffffu) {
    constexpr unsigned period = PER0 + PER1;
    const unsigned g = item / gsize, k = item % gsize, j = k / period, m = k % period;
    if (xs == 1u) nb = f.batch;
    FusedItem it;
    it.dep = nullptr;
    it.target = 0;
    if (m < PER0) {
        it.pass = g < nb ? 0u : 2u;
        it.t = x + xs * g;
        it.slot = x * f.ring * (xs >> 3) + g % f.ring;
        it.tile = j * PER0 + m;
        if (it.pass == 0u && g >= f.ring) {
            it.dep = rdone + kFusedCS * (x + xs * (g - f.ring));   // the ring slot it overwrites has been read
            it.target = f.tiles1;
        }
    } else {
        it.pass = g >= f.lag ? 1u : 2u;
        it.t = x + xs * (g - f.lag);
        it.slot = x * f.ring * (xs >> 3) + (g - f.lag) % f.ring;
        it.tile = j * PER1 + (m - PER0);
        if (it.pass == 1u) {
            it.dep = wdone + kFusedCS * it.t;           // every pass-0 tile of the transform has published
            it.target = f.tiles0;
        }
    }
    return it;
}

// lane 0 only: hand out this item's ticket and its early poll, draw the next ticket
// (tmul, tadd): a drawn ticket k is item k * tmul + tadd -- (1, 0) for one ticket counter; (8, x) when the counter is sharded per
// XCD and XCD x owns the items x, x + 8, ... of the SAME global list (development switch, see fused_loop)
__device__ __forceinline__ unsigned fused_advance(FusedQueue& q, unsigned& seen, unsigned total, unsigned* next, unsigned tmul = 1u,
                                                  unsigned tadd = 0u) {
    const unsigned item = q.t1;
    seen = q.seen1;
    q.seen1 = 0u;     // "not polled yet" (the hook of this item's tile sets it)
    if (item < total) q.t1 = __hip_atomic_fetch_add(next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * tmul + tadd;
    return item;
}

// The early poll.  A dependency is typically 20-50 us old when its consumer arrives (measured: 2 % younger than 10 us), so a
// poll from the middle of the previous tile (~8 us earlier) almost always sees it; one item earlier it fails three times in four.
template <unsigned PER0, unsigned PER1> struct FusedHook {
    const FusedArgs& f;
    FusedQueue& q;
    unsigned total, gsize;
    unsigned *wdone, *rdone;
    unsigned x, xs, nb;
    __device__ __forceinline__ void operator()() const {
        if (threadIdx.x == 0 && q.t1 < total) {
            const FusedItem nx = fused_decode<PER0, PER1>(f, q.t1, gsize, wdone, rdone, x, xs, nb);
            if (nx.dep != nullptr) q.seen1 = __hip_atomic_load(nx.dep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
};

// one persistent work-group: TILE0(t, slot, tile, hook) / TILE1(slot, t, tile, hook) run one tile of pass 0 / pass 1
// XCD = 1: one work list per XCD.  XCD = 2 (development): the ONE global list with its ticket counter sharded per XCD -- XCD x draws
// the items x, x + 8, ...; every dependency still points to a lower item of the same list, and the lowest unfinished item is
// either running or the next ticket of its XCD, so the order argument holds as long as every XCD has a resident work-group.
template <unsigned PER0, unsigned PER1, bool EARLY, typename TILE0, typename TILE1, int XCD = 0>
__device__ __forceinline__ void fused_loop(const FusedArgs& f, unsigned* s_item, TILE0&& tile0, TILE1&& tile1) {
    // XCD-local lists: ticket counter of XCD x on its own line behind the dependency counters, census word beside it
    const unsigned xq = XCD ? fused_xcc_id() : 0u;
    const unsigned x = XCD == 1 ? xq : 0u;
    constexpr unsigned xs = XCD == 1 ? 8u : 1u;
    const unsigned tmul = XCD == 2 ? 8u : 1u, tadd = XCD == 2 ? xq : 0u;
    const unsigned nb = XCD == 1 ? (f.batch + 7u - x) >> 3 : f.batch;
    unsigned* const next = XCD ? f.counters + kFusedCS * (1u + 2u * f.batch + xq) : f.counters;
    unsigned* const err = f.counters + 1;
    if constexpr (XCD) {
        if (threadIdx.x == 0) __hip_atomic_fetch_add(next + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // census
    }
    // One counter per 256-byte line: the counters of the few transforms in flight are polled and bumped by all 512 work-groups,
    // and packed 32 to a line they shared one memory channel's atomic unit (C2 with no polls at all -- wrong results, same
    // traffic -- ran 15 % faster; hiding the poll LATENCY changed nothing: it is the rate of same-line agent-scope accesses).
    unsigned* const wdone = f.counters + kFusedCS;
    unsigned* const rdone = wdone + kFusedCS * f.batch;
    // a group = the tiles0 pass-0 tiles of transform g and the tiles1 pass-1 tiles of transform g - lag, interleaved in their
    // ratio (tiles0 : tiles1 = PER0 : PER1), so that no ticket is an empty item
    const unsigned gsize = f.tiles0 + f.tiles1;
    const unsigned total = (nb + f.lag) * gsize;

    FusedPending pend = {nullptr};
    FusedQueue q = {0u, 0u};
    if (threadIdx.x == 0) q.t1 = __hip_atomic_fetch_add(next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * tmul + tadd;
    const FusedHook<PER0, PER1> hook = {f, q, total, gsize, wdone, rdone, x, xs, nb};
    for (;;) {
        __syncthreads();  // the previous item's LDS traffic and its s_item read are over
        unsigned seen = 0;
        if (threadIdx.x == 0) *s_item = fused_advance(q, seen, total, next, tmul, tadd);
        __syncthreads();
        const unsigned item = *s_item;
        if (item >= total) break;
        const FusedItem it = fused_decode<PER0, PER1>(f, item, gsize, wdone, rdone, x, xs, nb);
        if (it.pass == 2u) {             // fill / drain of the pipeline: nothing to do, but never sit on a publish
            fused_flush(pend);
            continue;
        }
        if (it.pass == 0u) {
            if (it.dep != nullptr) fused_wait_ge<false>(it.dep, it.target, seen, err, pend);
            else fused_flush(pend);
            if constexpr (EARLY) tile0(it.t, it.slot, it.tile, hook);
            else tile0(it.t, it.slot, it.tile, TileNoHook());
            pend.ctr = wdone + kFusedCS * it.t;     // published behind the next item's dependency wait (or at the end)
        } else {
            // (reading the ring with sc1 loads instead of the acquire fence measured the same, and 8-byte sc1 loads at two
            // work-groups per CU are outside the hand-off forms MI355X_MICROARCH.md lists as validated: the fence stays)
            fused_wait_ge<true>(it.dep, it.target, seen, err, pend);
            if constexpr (EARLY) tile1(it.slot, it.t, it.tile, hook);
            else tile1(it.slot, it.t, it.tile, TileNoHook());
            fused_signal_read(rdone + kFusedCS * it.t);
        }
    }
    fused_flush(pend);
}

// NT: 0 = plain accesses on the streamed side, 1 = non-temporal loads of the input and stores of the output, 2 = non-temporal
// loads and write-through (sc1) stores of the output (A/B; non-temporal loads of the RING in pass 1 measured 0.6 % slower on C2)
template <typename T, int A0, int A1, bool SPLIT, int NT>
__global__ void __launch_bounds__(256, 2) fft_fused2_kernel(const FusedArgs f) {
    constexpr int E0 = Col2Lds<A0, true>::ELEMS, E1 = Col2Lds<A1, false>::ELEMS;
    __shared__ __attribute__((aligned(16))) cplx<T> lds[E0 > E1 ? E0 : E1];
    __shared__ unsigned s_item;
    constexpr unsigned per0 = A0 > A1 ? 1u : (unsigned)(A1 / A0);   // tiles0 : tiles1 = L1 : L0 = A1 : A0
    constexpr unsigned per1 = A1 > A0 ? 1u : (unsigned)(A0 / A1);
    fused_loop<per0, per1, !SPLIT>(   // early poll: C2 19.49 -> 19.32 ms; split planes 26.7 -> 25.6 ms WITHOUT it
        f, &s_item,
        [&](unsigned t, unsigned slot, unsigned tile, auto hook) {
            col2_tile<T, A0, true, true, SPLIT, true, NT != 0, false, false>(f.p0, (long long)t, (long long)slot, (long long)tile * 16, lds, hook);
        },
        [&](unsigned slot, unsigned t, unsigned tile, auto hook) {
            col2_tile<T, A1, false, false, false, NT == 2, false, NT == 1, SPLIT>(f.p1, (long long)slot, (long long)t, (long long)tile * 16, lds, hook);
        });
}

// XCD-local lists (see FusedArgs above).  WT0: write the intermediate with write-through stores (as the global form must) or
// with plain ones (it then lives in the XCD's L2 and is written back only when evicted).
template <typename T, int A0, int A1, bool WT0, int XCD = 1>
__global__ void __launch_bounds__(256, 2) fft_fused2x_kernel(const FusedArgs f) {
    constexpr int E0 = Col2Lds<A0, true>::ELEMS, E1 = Col2Lds<A1, false>::ELEMS;
    __shared__ __attribute__((aligned(16))) cplx<T> lds[E0 > E1 ? E0 : E1];
    __shared__ unsigned s_item;
    constexpr unsigned per0 = A0 > A1 ? 1u : (unsigned)(A1 / A0);
    constexpr unsigned per1 = A1 > A0 ? 1u : (unsigned)(A0 / A1);
    auto t0 = [&](unsigned t, unsigned slot, unsigned tile, auto hook) {
        col2_tile<T, A0, true, true, false, WT0, true, false, false>(f.p0, (long long)t, (long long)slot, (long long)tile * 16, lds, hook);
    };
    auto t1 = [&](unsigned slot, unsigned t, unsigned tile, auto hook) {
        col2_tile<T, A1, false, false, false, false, false, true, false>(f.p1, (long long)slot, (long long)t, (long long)tile * 16, lds, hook);
    };
    fused_loop<per0, per1, true, decltype(t0)&, decltype(t1)&, XCD>(f, &s_item, t0, t1);
}

// The 2-D form (BASELINE config 3: 1024 x 1024): a 2-D transform is two TRANSPOSING column passes without an inter-pass twiddle
// -- pass 0 transforms the y axis of in[y][x] and writes ring[x][ky], pass 1 transforms the x axis of that and writes
// out[ky][kx] -- i.e. the 1-D kernel above minus the twiddle, with contiguous 8 KiB runs on the output side.
template <typename T, int A, bool SPLIT, bool NT>
__global__ void __launch_bounds__(256, 2) fft_fused2d_kernel(const FusedArgs f) {
    __shared__ __attribute__((aligned(16))) cplx<T> lds[Col2Lds<A, true>::ELEMS];
    __shared__ unsigned s_item;
    fused_loop<1, 1, !SPLIT>(
        f, &s_item,
        [&](unsigned t, unsigned slot, unsigned tile, auto hook) {
            col2_tile<T, A, true, false, SPLIT, true, NT, false, false>(f.p0, (long long)t, (long long)slot, (long long)tile * 16, lds, hook);
        },
        [&](unsigned slot, unsigned t, unsigned tile, auto hook) {
            col2_tile<T, A, true, false, false, false, false, NT, SPLIT>(f.p1, (long long)slot, (long long)t, (long long)tile * 16, lds, hook);
        });
}

// The same work list with the 512-thread tiles of fft_col3.hpp (L = 512 * A): fp32 N = 2^22 = 2048 x 2048 (BASELINE config 5;
// one work-group per CU, the ring holds 7 transforms of 32 MiB), fp32 N = 2^21 = 2048 x 1024 and fp64 N = 2^20 = 1024 x 1024
// (14 transforms of 16 MiB).
template <typename T, int A0, int A1, bool SPLIT, bool NT>
__global__ void __launch_bounds__(512, 2) fft_fused3_kernel(const FusedArgs f) {
    constexpr int E0 = Col3Lds<T, true>::SCALARS, E1 = Col3Lds<T, false>::SCALARS;
    __shared__ __attribute__((aligned(16))) T lds[E0 > E1 ? E0 : E1];
    __shared__ unsigned s_item;
    constexpr unsigned per0 = A0 > A1 ? 1u : (unsigned)(A1 / A0);
    constexpr unsigned per1 = A1 > A0 ? 1u : (unsigned)(A0 / A1);
    fused_loop<per0, per1, false>(   // no early poll: the fp64 tiles have no register to spare for it, the fp32 ones measured equal
        f, &s_item,
        [&](unsigned t, unsigned slot, unsigned tile, auto hook) {
            col3_tile<T, A0, true, true, SPLIT, NT, false, false, true>(f.p0, (long long)t, (long long)slot, (long long)tile * 16, lds, hook);
        },
        [&](unsigned slot, unsigned t, unsigned tile, auto hook) {
            col3_tile<T, A1, false, false, false, false, NT, SPLIT>(f.p1, (long long)slot, (long long)t, (long long)tile * 16, lds, hook);
        });
}

// 2-D 1024 x 1024 in fp64 on the 512-thread tiles (the published (1024, 1024) double-precision shape): fft_fused2d_kernel's data flow
template <typename T, int A, bool SPLIT, bool NT>
__global__ void __launch_bounds__(512, 2) fft_fused3d_kernel(const FusedArgs f) {
    __shared__ __attribute__((aligned(16))) T lds[Col3Lds<T, true>::SCALARS];
    __shared__ unsigned s_item;
    fused_loop<1, 1, false>(
        f, &s_item,
        [&](unsigned t, unsigned slot, unsigned tile, auto hook) {
            col3_tile<T, A, true, false, SPLIT, NT, false, false, true>(f.p0, (long long)t, (long long)slot, (long long)tile * 16, lds, hook);
        },
        [&](unsigned slot, unsigned t, unsigned tile, auto hook) {
            col3_tile<T, A, true, false, false, false, NT, SPLIT>(f.p1, (long long)slot, (long long)t, (long long)tile * 16, lds, hook);
        });
}

}  // namespace mifft

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
#include "fft_col2w.hpp"
#include "fft_col3.hpp"

namespace mifft {

constexpr unsigned kFusedCS = 64u;   // words between two counters (= MIFFT_FUSED2_COUNTER_STRIDE)

// the work list and its synchronisation state (shared by every persistent kernel: fft_fused2 / fused3 / fused2x / fusedp)
struct FusedCtl {
    unsigned* counters;  // [0] ticket, [1] error (when `err` points here), then one counter per CS = 64 words (256 bytes): wdone[t] at
                         // CS * (1 + t), rdone[t] at CS * (1 + batch + t), the ticket counters of the per-XCD lists behind them; ALL ZERO
                         // when the launch starts
    unsigned* counters_next;   // nullptr, or a second counter set that THIS launch zeroes for the next one (round 4: a plan alternates
                               // between two sets, so no memset node precedes the launch)
    unsigned* err;       // error word, set on a dependency time-out (device-accessible: counters + 1, or pinned host memory)
    unsigned lines;      // counter lines per set (9 + 2 * batch)
    unsigned batch;      // number of transforms
    unsigned lag;        // pass 1 of transform t is queued with pass 0 of transform t + lag; 0 = the SEQUENTIAL list of a tiny batch:
                         // every pass-0 tile of every transform, then every pass-1 tile (ring = batch slots, no slot is reused)
    unsigned ring;       // scratch ring slots (transforms); ring > lag
    unsigned tiles0;     // tiles per transform in pass 0
    unsigned tiles1;     // tiles per transform in pass 1
};

struct FusedArgs {
    TileArgs p0;         // pass 0: in = user input,  out = scratch ring (matrix index = ring slot)
    TileArgs p1;         // pass 1: in = scratch ring, out = user output
    FusedCtl c;
};

// XCD-local form (strategy `fusedx`, round 3; a default since round 4): one work list PER XCD (chiplet).  A work-group reads its XCD
// from HW_REG_XCC_ID and draws tickets from that XCD's counter; XCD x owns the transforms t = x + 8 i, its ring slots are
// [x * ring, (x + 1) * ring).  (Round 3 measured that the intermediate still crosses the fabric twice -- an XCD's L2 keeps ~1 MiB
// next to the streams, a stall-free list needs 4-32 MiB -- so what the form buys is eight short pipelines with eight ticket
// counters instead of one long one: + 2 points at 2^16 / 2^17, profiles/r04_a_fused_sweep.log.)  Same dependency order per list, so
// the same no-deadlock argument.  Round 4: a work-group that finds its own list exhausted moves on to the lists other XCDs have
// not finished (work stealing, in list order), so every list is drained whatever the placement of the work-groups -- a launch
// that leaves an XCD without work-groups (a CU mask, a shared device) is slower, not wrong.
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
#ifdef MIFFT_DEV_BUILD
    unsigned max_spins;   // `make DEV=1`: the longest dependency wait of this work-group in polls (tools/spin_margin.py)
#endif
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
                __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (may be pinned host memory)
                break;
            }
        }
#ifdef MIFFT_DEV_BUILD
        if (spins > pend.max_spins) pend.max_spins = spins;
#endif
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
__device__ __forceinline__ FusedItem fused_decode(const FusedCtl& f, unsigned item, unsigned gsize, unsigned* wdone, unsigned* rdone,
                                                  unsigned x = 0u, unsigned xs = 1u, unsigned nb = 0xffffffffu) {
    FusedItem it;
    it.dep = nullptr;
    it.target = 0;
    if (f.lag == 0u) {
        // sequential list (tiny batches): no empty item, no ring reuse; a pass-1 tile still waits for its transform's pass-0 tiles
        const unsigned n0 = f.batch * f.tiles0;
        if (item < n0) {
            it.pass = 0u;
            it.t = item / f.tiles0;
            it.tile = item % f.tiles0;
        } else {
            it.pass = 1u;
            it.t = (item - n0) / f.tiles1;
            it.tile = (item - n0) % f.tiles1;
            it.dep = wdone + kFusedCS * it.t;
            it.target = f.tiles0;
        }
        it.slot = it.t;
        return it;
    }
    constexpr unsigned period = PER0 + PER1;
    const unsigned g = item / gsize, k = item % gsize, j = k / period, m = k % period;
    if (xs == 1u) nb = f.batch;
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

// lane 0 only: hand out this item's ticket and its early poll, draw the next ticket.  `stat`: the sequential list of a tiny batch is
// dealt out STATICALLY (work-group w takes the items w, w + grid, ...): with every work-group of the launch resident (grid <= 2 per
// CU, checked by the host) nothing can deadlock -- a work-group runs all its first-pass items, which never wait, before its first
// second-pass item -- and 512 work-groups do not queue up twice at one ticket counter (measured: 53 against 25 us for two plain
// launches at (1024, 1024) x 4 with tickets, profiles/r04_b_small_batch_sequential.log).
__device__ __forceinline__ unsigned fused_advance(FusedQueue& q, unsigned& seen, unsigned total, unsigned* next, bool stat) {
    const unsigned item = q.t1;
    seen = q.seen1;
    q.seen1 = 0u;     // "not polled yet" (the hook of this item's tile sets it)
    if (item < total) q.t1 = stat ? item + gridDim.x : __hip_atomic_fetch_add(next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return item;
}

// The early poll.  A dependency is typically 20-50 us old when its consumer arrives (measured: 2 % younger than 10 us), so a
// poll from the middle of the previous tile (~8 us earlier) almost always sees it; one item earlier it fails three times in four.
template <unsigned PER0, unsigned PER1> struct FusedHook {
    const FusedCtl& f;
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

// one work list (the global one, or the list of XCD x), drained by this work-group together with whoever else draws from `next`
template <unsigned PER0, unsigned PER1, bool EARLY, typename TILE0, typename TILE1>
__device__ __forceinline__ void fused_list(const FusedCtl& f, unsigned* s_item, TILE0& tile0, TILE1& tile1, FusedPending& pend,
                                           const unsigned x, const unsigned xs, const unsigned nb, unsigned* const next) {
    unsigned* const err = f.err;
    // One counter per 256-byte line: the counters of the few transforms in flight are polled and bumped by all 512 work-groups,
    // and packed 32 to a line they shared one memory channel's atomic unit (C2 with no polls at all -- wrong results, same
    // traffic -- ran 15 % faster; hiding the poll LATENCY changed nothing: it is the rate of same-line agent-scope accesses).
    unsigned* const wdone = f.counters + kFusedCS;
    unsigned* const rdone = wdone + kFusedCS * f.batch;
    // a group = the tiles0 pass-0 tiles of transform g and the tiles1 pass-1 tiles of transform g - lag, interleaved in their
    // ratio (tiles0 : tiles1 = PER0 : PER1), so that no ticket is an empty item
    const unsigned gsize = f.tiles0 + f.tiles1;
    const unsigned total = f.lag == 0u ? f.batch * gsize : (nb + f.lag) * gsize;

    const bool stat = f.lag == 0u;
    FusedQueue q = {0u, 0u};
    if (threadIdx.x == 0) q.t1 = stat ? blockIdx.x : __hip_atomic_fetch_add(next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const FusedHook<PER0, PER1> hook = {f, q, total, gsize, wdone, rdone, x, xs, nb};
    for (;;) {
        __syncthreads();  // the previous item's LDS traffic and its s_item read are over
        unsigned seen = 0;
        if (threadIdx.x == 0) *s_item = fused_advance(q, seen, total, next, stat);
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
    fused_flush(pend);   // never carry a publish into another list (or out of the kernel)
}

// one persistent work-group: TILE0(t, slot, tile, hook) / TILE1(slot, t, tile, hook) run one tile of pass 0 / pass 1
// XCD = 1: one work list per XCD, own list first, then the unfinished lists of the other XCDs.
template <unsigned PER0, unsigned PER1, bool EARLY, typename TILE0, typename TILE1, int XCD = 0>
__device__ __forceinline__ void fused_loop(const FusedCtl& f, unsigned* s_item, TILE0&& tile0, TILE1&& tile1) {
    // the counter set of the NEXT launch (words 0 and 1 of every line: a counter, or ticket + error / census)
    if (f.counters_next != nullptr) {
        for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < f.lines; i += gridDim.x * blockDim.x) {
            f.counters_next[i * kFusedCS] = 0u;
            f.counters_next[i * kFusedCS + 1u] = 0u;
        }
    }
    FusedPending pend = {};
    if constexpr (XCD == 0) {
        fused_list<PER0, PER1, EARLY>(f, s_item, tile0, tile1, pend, 0u, 1u, f.batch, f.counters);
#ifdef MIFFT_DEV_BUILD
        // word 2 of the ticket line: the longest dependency wait of the launches on this counter set, in polls of ~1 us (the time-out is
        // 2^22); nothing else writes it, and only a memset of the set clears it
        if (threadIdx.x == 0 && pend.max_spins) __hip_atomic_fetch_max(f.counters + 2, pend.max_spins, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    } else {
        // ticket counter of XCD x on its own line behind the dependency counters
        const unsigned home = fused_xcc_id();
        unsigned* const tickets = f.counters + kFusedCS * (1u + 2u * f.batch);
        const unsigned gsize = f.tiles0 + f.tiles1;
        unsigned visit = 1u;     // bit h: drain the list of XCD (home + h) & 7
        for (unsigned h = 0; h < 8u; ++h) {
            if (visit & (1u << h)) {
                const unsigned x = (home + h) & 7u;
                fused_list<PER0, PER1, EARLY>(f, s_item, tile0, tile1, pend, x, 8u, (f.batch + 7u - x) >> 3, tickets + kFusedCS * x);
            }
            if (h == 0u) {
                // own list exhausted: which other lists still have tickets?  (seven loads in flight at once, one lane)
                __syncthreads();
                if (threadIdx.x == 0) {
                    unsigned v[7];
#pragma unroll
                    for (unsigned k = 0; k < 7u; ++k)
                        v[k] = __hip_atomic_load(tickets + kFusedCS * ((home + 1u + k) & 7u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    unsigned mask = 1u;
#pragma unroll
                    for (unsigned k = 0; k < 7u; ++k) {
                        const unsigned x = (home + 1u + k) & 7u;
                        if (v[k] < (((f.batch + 7u - x) >> 3) + f.lag) * gsize) mask |= 2u << k;
                    }
                    *s_item = mask;
                }
                __syncthreads();
                visit = *s_item;
            }
        }
    }
}

// NT: 0 = plain accesses on the streamed side, 1 = non-temporal loads of the input and stores of the output, 2 = non-temporal
// loads and write-through (sc1) stores of the output (A/B; non-temporal loads of the RING in pass 1 measured 0.6 % slower on C2)
template <typename T, int A0, int A1, bool SPLIT, int NT>
__global__ void __launch_bounds__(256, 2) fft_fused2_kernel(const FusedArgs f) {
    constexpr int E0 = Col2Lds<A0, true, sizeof(cplx<T>)>::ELEMS, E1 = Col2Lds<A1, false, sizeof(cplx<T>)>::ELEMS;
    __shared__ __attribute__((aligned(16))) cplx<T> lds[E0 > E1 ? E0 : E1];
    __shared__ unsigned s_item;
    constexpr unsigned per0 = A0 > A1 ? 1u : (unsigned)(A1 / A0);   // tiles0 : tiles1 = L1 : L0 = A1 : A0
    constexpr unsigned per1 = A1 > A0 ? 1u : (unsigned)(A0 / A1);
    fused_loop<per0, per1, !SPLIT && sizeof(T) == 4>(   // early poll: C2 19.49 -> 19.32 ms; split planes 26.7 -> 25.6 ms WITHOUT it; fp64: no register to spare
        f.c, &s_item,
        [&](unsigned t, unsigned slot, unsigned tile, auto hook) {
            col2_tile<T, A0, true, true, SPLIT, true, NT != 0, false, false>(f.p0, (long long)t, (long long)slot, (long long)tile * 16, lds, hook);
        },
        [&](unsigned slot, unsigned t, unsigned tile, auto hook) {
            col2_tile<T, A1, false, false, false, NT == 2, false, NT == 1, SPLIT>(f.p1, (long long)slot, (long long)t, (long long)tile * 16, lds, hook);
        });
}

// Split-complex fp32 with the SIBLING tiles side by side (second batch of round 4).  A 16-column tile touches 64 bytes per row of
// each plane -- half of a 128-byte line; on the global list the tile that owns the other half runs on whatever XCD drew its ticket
// and every input line crosses the fabric twice (PMC 2.48 x the algorithmic bytes, profiles/r04v_pmc_traffic_split.log).  Here a
// work-group is 512 threads = two 256-thread tiles in lock step (same code, same barriers), each on its own half of the LDS array,
// and an item is the sibling pair: both halves of every line are requested by the same CU at the same time.  One work-group per CU
// (the two tiles take the registers of two ordinary work-groups).  TWOD: the 2-D data flow (two transposing passes, no twiddle).
template <int A0, int A1, bool TWOD, bool NT = false, bool SPLIT = true>
__global__ void __launch_bounds__(512, 2) fft_fused2s_kernel(const FusedArgs f) {
    constexpr int E0 = Col2Lds<A0, true, sizeof(cplx<float>)>::ELEMS, E1 = Col2Lds<A1, TWOD, sizeof(cplx<float>)>::ELEMS;
    constexpr int E = E0 > E1 ? E0 : E1;
    __shared__ __attribute__((aligned(16))) cplx<float> lds[2 * E];
    __shared__ unsigned s_item;
    constexpr unsigned per0 = A0 > A1 ? 1u : (unsigned)(A1 / A0);
    constexpr unsigned per1 = A1 > A0 ? 1u : (unsigned)(A0 / A1);
    fused_loop<per0, per1, false>(
        f.c, &s_item,
        [&](unsigned t, unsigned slot, unsigned item, auto hook) {
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));
            col2_tile<float, A0, true, !TWOD, SPLIT, true, NT, false, false, true>(f.p0, (long long)t, (long long)slot, (long long)item * 32, lds, hook, tid);
        },
        [&](unsigned slot, unsigned t, unsigned item, auto hook) {
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));
            col2_tile<float, A1, TWOD, false, false, false, false, NT, SPLIT, true>(f.p1, (long long)slot, (long long)t, (long long)item * 32, lds, hook, tid);
        });
}

// XCD-local lists (see fused_xcc_id above).  The intermediate is written WRITE-THROUGH, as in the global form: a work-group that
// drains another XCD's list (work stealing) produces and consumes across XCDs, and only write-through stores + the consumer's
// acquire are coherent between two L2s.  (Round 3's plain-store variant -- the intermediate in the owning XCD's L2 -- measured
// 0-2 points more and was placement-dependent: removed.)
template <typename T, int A0, int A1, bool SPLIT = false>
__global__ void __launch_bounds__(256, 2) fft_fused2x_kernel(const FusedArgs f) {
    constexpr int E0 = Col2Lds<A0, true, sizeof(cplx<T>)>::ELEMS, E1 = Col2Lds<A1, false, sizeof(cplx<T>)>::ELEMS;
    __shared__ __attribute__((aligned(16))) cplx<T> lds[E0 > E1 ? E0 : E1];
    __shared__ unsigned s_item;
    constexpr unsigned per0 = A0 > A1 ? 1u : (unsigned)(A1 / A0);
    constexpr unsigned per1 = A1 > A0 ? 1u : (unsigned)(A0 / A1);
    auto t0 = [&](unsigned t, unsigned slot, unsigned tile, auto hook) {
        col2_tile<T, A0, true, true, SPLIT, true, !SPLIT, false, false>(f.p0, (long long)t, (long long)slot, (long long)tile * 16, lds, hook);
    };
    auto t1 = [&](unsigned slot, unsigned t, unsigned tile, auto hook) {
        col2_tile<T, A1, false, false, false, false, false, !SPLIT, SPLIT>(f.p1, (long long)slot, (long long)t, (long long)tile * 16, lds, hook);
    };
    fused_loop<per0, per1, !SPLIT, decltype(t0)&, decltype(t1)&, 1>(f.c, &s_item, t0, t1);
}

// The 1-D kernel on the 32-column tiles of fft_col2w.hpp (round 4): fp32 interleaved, L0 >= L1 in {256, 512} (N = 2^16 ... 2^18);
// a tile is 32 columns, so a transform has L1 / 32 first-pass and L0 / 32 second-pass tiles.
template <int A0, int A1>
__global__ void __launch_bounds__(256, 2) fft_fused2w_kernel(const FusedArgs f) {
    constexpr int E0 = Col2wLds<true>::ELEMS, E1 = Col2wLds<false>::ELEMS;
    __shared__ __attribute__((aligned(16))) cplx<float> lds[E0 > E1 ? E0 : E1];
    __shared__ unsigned s_item;
    constexpr unsigned per0 = A0 > A1 ? 1u : (unsigned)(A1 / A0);
    constexpr unsigned per1 = A1 > A0 ? 1u : (unsigned)(A0 / A1);
    fused_loop<per0, per1, (A0 == 1)>(     // (the early dependency poll only where registers are to spare: L = 512 sits at the 256-VGPR limit)
        f.c, &s_item,
        [&](unsigned t, unsigned slot, unsigned tile, auto hook) {
            col2w_tile<A0, true, true, true, true, false>(f.p0, (long long)t, (long long)slot, (long long)tile * 32, lds, hook);
        },
        [&](unsigned slot, unsigned t, unsigned tile, auto hook) {
            col2w_tile<A1, false, false, false, false, true>(f.p1, (long long)slot, (long long)t, (long long)tile * 32, lds, hook);
        });
}

// N = 2^19 = 1024 x 512 (round 4): the 1024-point pass on the 16-column tiles (64 points per thread: no room for a second column),
// the 512-point pass on the 32-column ones -- 32 + 32 tiles per transform
template <int NT>
__global__ void __launch_bounds__(256, 2) fft_fused2m_kernel(const FusedArgs f) {
    constexpr int E0 = Col2Lds<4, true>::ELEMS, E1 = Col2wLds<false>::ELEMS;
    __shared__ __attribute__((aligned(16))) cplx<float> lds[E0 > E1 ? E0 : E1];
    __shared__ unsigned s_item;
    fused_loop<1, 1, false>(
        f.c, &s_item,
        [&](unsigned t, unsigned slot, unsigned tile, auto hook) {
            col2_tile<float, 4, true, true, false, true, NT != 0, false, false>(f.p0, (long long)t, (long long)slot, (long long)tile * 16, lds, hook);
        },
        [&](unsigned slot, unsigned t, unsigned tile, auto hook) {
            col2w_tile<2, false, false, false, false, NT != 0>(f.p1, (long long)slot, (long long)t, (long long)tile * 32, lds, hook);
        });
}

// The 2-D form (BASELINE config 3: 1024 x 1024): a 2-D transform is two TRANSPOSING column passes without an inter-pass twiddle
// -- pass 0 transforms the y axis of in[y][x] and writes ring[x][ky], pass 1 transforms the x axis of that and writes
// out[ky][kx] -- i.e. the 1-D kernel above minus the twiddle, with contiguous 8 KiB runs on the output side.
// Round 4: rectangles.  A0 = ny / 256 (pass 0 transforms the y axis over nx / 16 tiles), A1 = nx / 256 (pass 1 the x axis over
// ny / 16 tiles); the two passes look their w(L) up in different tables (TileArgs.tw_L of p0 / p1).
template <typename T, int A0, int A1, bool SPLIT, bool NT>
__global__ void __launch_bounds__(256, 2) fft_fused2d_kernel(const FusedArgs f) {
    constexpr int E0 = Col2Lds<A0, true, sizeof(cplx<T>)>::ELEMS, E1 = Col2Lds<A1, true, sizeof(cplx<T>)>::ELEMS;
    __shared__ __attribute__((aligned(16))) cplx<T> lds[E0 > E1 ? E0 : E1];
    __shared__ unsigned s_item;
    constexpr unsigned per0 = A0 > A1 ? 1u : (unsigned)(A1 / A0);   // tiles0 : tiles1 = nx : ny = A1 : A0
    constexpr unsigned per1 = A1 > A0 ? 1u : (unsigned)(A0 / A1);
    fused_loop<per0, per1, !SPLIT && sizeof(T) == 4>(
        f.c, &s_item,
        [&](unsigned t, unsigned slot, unsigned tile, auto hook) {
            col2_tile<T, A0, true, false, SPLIT, true, NT, false, false>(f.p0, (long long)t, (long long)slot, (long long)tile * 16, lds, hook);
        },
        [&](unsigned slot, unsigned t, unsigned tile, auto hook) {
            col2_tile<T, A1, true, false, false, false, false, NT, SPLIT>(f.p1, (long long)slot, (long long)t, (long long)tile * 16, lds, hook);
        });
}

// The 2-D form with 32-column tiles for a 512-point axis (round 4): (512, 512), (512, 1024), (1024, 512) fp32 interleaved.  W0 / W1 =
// columns per tile of pass 0 / pass 1: 32 where the pass transforms 512 points, 16 where it transforms 1024.
template <int A0, int A1>
__global__ void __launch_bounds__(256, 2) fft_fused2dw_kernel(const FusedArgs f) {
    constexpr bool WIDE0 = A0 <= 2, WIDE1 = A1 <= 2;
    constexpr int E0 = WIDE0 ? Col2wLds<true>::ELEMS : Col2Lds<A0, true>::ELEMS, E1 = WIDE1 ? Col2wLds<true>::ELEMS : Col2Lds<A1, true>::ELEMS;
    __shared__ __attribute__((aligned(16))) cplx<float> lds[E0 > E1 ? E0 : E1];
    __shared__ unsigned s_item;
    // tiles0 : tiles1 = nx / W0 : ny / W1 with nx = 256 A1, ny = 256 A0
    constexpr unsigned t0 = 256u * A1 / (WIDE0 ? 32u : 16u), t1 = 256u * A0 / (WIDE1 ? 32u : 16u);
    constexpr unsigned per0 = t0 >= t1 ? t0 / t1 : 1u, per1 = t1 > t0 ? t1 / t0 : 1u;
    fused_loop<per0, per1, false>(
        f.c, &s_item,
        [&](unsigned t, unsigned slot, unsigned tile, auto hook) {
            if constexpr (WIDE0) col2w_tile<A0, true, false, true, true, false>(f.p0, (long long)t, (long long)slot, (long long)tile * 32, lds, hook);
            else col2_tile<float, A0, true, false, false, true, true, false, false>(f.p0, (long long)t, (long long)slot, (long long)tile * 16, lds, hook);
        },
        [&](unsigned slot, unsigned t, unsigned tile, auto hook) {
            if constexpr (WIDE1) col2w_tile<A1, true, false, false, false, true>(f.p1, (long long)slot, (long long)t, (long long)tile * 32, lds, hook);
            else col2_tile<float, A1, true, false, false, false, false, true, false>(f.p1, (long long)slot, (long long)t, (long long)tile * 16, lds, hook);
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
        f.c, &s_item,
        [&](unsigned t, unsigned slot, unsigned tile, auto hook) {
            col3_tile<T, A0, true, true, SPLIT, NT, false, false, true>(f.p0, (long long)t, (long long)slot, (long long)tile * 16, lds, hook);
        },
        [&](unsigned slot, unsigned t, unsigned tile, auto hook) {
            col3_tile<T, A1, false, false, false, false, NT, SPLIT>(f.p1, (long long)slot, (long long)t, (long long)tile * 16, lds, hook);
        });
}

// ---- round 6: the next tile's loads issued at the END of the current tile ---------------------------------------------------------------
// With one 512-thread work-group per CU nothing overlaps a tile's dead time: at the top of every item the work-group drains its stores
// (publish), polls the item's dependency (an agent-scope load: 1-3 us), runs an acquire, and only then issues the tile's loads and waits
// a memory latency for the first of them -- 4-5 us of a 21 us tile in which this CU moves no data (profiles/r06_fused3_counters.log:
// 102 read requests in flight per L2 channel against 127 for the 256-thread kernel at two work-groups per CU).  Here thread 0 hands
// out the NEXT item from inside the current tile's last exchange round (hook2: the round's own barrier publishes it to the work-group),
// and the loop issues that item's loads FIRST, right behind the previous tile's last stores -- before the publish (store drain +
// barrier + counter), which then runs concurrently with the load latency; no barrier and no LDS round trip lie between two tiles.  A second-pass item is prefetched only if the early poll of its
// dependency (the hook in the middle of the tile) has already seen it satisfied -- thread 0 then runs the acquire in hook2, in front
// of the barrier; otherwise the item takes the ordinary path at the top of the loop.  First-pass items read the user's input, which
// depends on nothing (their dependency guards the ring slot they WRITE, and is still waited for before the tile computes).
// Tickets stay global and increasing, every work-group still works through its tickets in order and publishes before it waits: the
// no-deadlock argument of the list above is unchanged.  LOAD0(t, tile) / LOAD1(slot, tile) issue a tile's loads into the caller's registers.
template <unsigned PER0, unsigned PER1, typename LOAD0, typename LOAD1, typename TILE0, typename TILE1>
__device__ __forceinline__ void fused_list_prefetch(const FusedCtl& f, unsigned* s_item, LOAD0& load0, LOAD1& load1, TILE0& tile0, TILE1& tile1) {
    unsigned* const err = f.err;
    unsigned* const next = f.counters;
    unsigned* const wdone = f.counters + kFusedCS;
    unsigned* const rdone = wdone + kFusedCS * f.batch;
    const unsigned gsize = f.tiles0 + f.tiles1;
    const unsigned total = (f.batch + f.lag) * gsize;
    FusedPending pend = {nullptr};
    FusedQueue q = {0u, 0u};
    unsigned seen_cur = 0;       // thread 0: the early poll of the CURRENT item's dependency
    if (threadIdx.x == 0) {
        s_item[0] = __hip_atomic_fetch_add(next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_item[1] = 0u;
        q.t1 = __hip_atomic_fetch_add(next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const FusedHook<PER0, PER1> hook = {f, q, total, gsize, wdone, rdone, 0u, 1u, f.batch};
    // thread 0: hand out the next item (s_item[0]) and say whether its loads may be issued at once (s_item[1])
    auto handout = [&]() {
        if (threadIdx.x == 0) {
            const unsigned nxt = q.t1;
            unsigned ok = 0u;
            seen_cur = q.seen1;
            q.seen1 = 0u;
            if (nxt < total) {
                const FusedItem nx = fused_decode<PER0, PER1>(f, nxt, gsize, wdone, rdone);
                if (nx.pass == 0u) {
                    ok = 1u;
                } else if (nx.pass == 1u && seen_cur >= nx.target) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    ok = 1u;
                }
                q.t1 = __hip_atomic_fetch_add(next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            s_item[0] = nxt;
            s_item[1] = ok;
        }
    };
    for (;;) {
        // (what thread 0 wrote in the previous tile's last round -- or in the prologue -- is visible: a barrier lies in between)
        const unsigned item = __builtin_amdgcn_readfirstlane(s_item[0]);
        const bool ok = __builtin_amdgcn_readfirstlane(s_item[1]) != 0u;
        if (item >= total) break;
        const FusedItem it = fused_decode<PER0, PER1>(f, item, gsize, wdone, rdone);
        if (it.pass == 0u) {
            load0(it.t, it.tile);                    // right behind the previous tile's last stores: the input depends on nothing
            if (it.dep != nullptr) fused_wait_ge<false>(it.dep, it.target, seen_cur, err, pend);    // (the ring slot it WRITES is free)
            else fused_flush(pend);
            tile0(it.t, it.slot, it.tile, hook, handout);
            pend.ctr = wdone + kFusedCS * it.t;     // published behind the next item's loads (or at the end)
        } else if (it.pass == 1u) {
            // ok: the early poll saw the dependency satisfied and thread 0 ran the acquire in front of the barrier that published this
            // item; else the ordinary path -- publish what this work-group owes, THEN wait
            if (!ok) fused_wait_ge<true>(it.dep, it.target, seen_cur, err, pend);
            load1(it.slot, it.tile);
            fused_flush(pend);
            tile1(it.slot, it.t, it.tile, hook, handout);
            fused_signal_read(rdone + kFusedCS * it.t);
        } else {                          // fill / drain of the pipeline: nothing to do, but never sit on a publish
            fused_flush(pend);
            __syncthreads();             // (the previous hand-out has been read by everybody)
            handout();
            __syncthreads();
        }
    }
    fused_flush(pend);
}

// fft_fused3_kernel with the prefetching list (interleaved data; A0 >= A1: the registers hold the larger tile)
template <typename T, int A0, int A1, bool NT>
__global__ void __launch_bounds__(512, 2) fft_fused3p_kernel(const FusedArgs f) {
    constexpr int E0 = Col3Lds<T, true>::SCALARS, E1 = Col3Lds<T, false>::SCALARS;
    __shared__ __attribute__((aligned(16))) T lds[E0 > E1 ? E0 : E1];
    __shared__ unsigned s_item[2];
    constexpr unsigned per0 = A0 > A1 ? 1u : (unsigned)(A1 / A0);
    constexpr unsigned per1 = A1 > A0 ? 1u : (unsigned)(A0 / A1);
    cplx<T> v[16 * (A0 > A1 ? A0 : A1)];
    if (f.c.counters_next != nullptr) {      // the counter set of the NEXT launch (as fused_loop)
        for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < f.c.lines; i += gridDim.x * blockDim.x) {
            f.c.counters_next[i * kFusedCS] = 0u;
            f.c.counters_next[i * kFusedCS + 1u] = 0u;
        }
    }
    Col3StageTw<T, A0> st0;
    Col3StageTw<T, A1> st1;
    auto l0 = [&](unsigned t, unsigned tile) {
        col3_load<T, A0, false, NT>(f.p0, (long long)t, (long long)tile * 16, v);
        st0.load(f.p0);
    };
    auto l1 = [&](unsigned slot, unsigned tile) {
        col3_load<T, A1, false, false>(f.p1, (long long)slot, (long long)tile * 16, v);
        st1.load(f.p1);
    };
    auto t0 = [&](unsigned t, unsigned slot, unsigned tile, auto hook, auto hook2) {
        col3_body<T, A0, true, true, false, false, true>(f.p0, (long long)slot, (long long)tile * 16, lds, v, st0, hook, hook2);
    };
    auto t1 = [&](unsigned slot, unsigned t, unsigned tile, auto hook, auto hook2) {
        col3_body<T, A1, false, false, NT, false, false>(f.p1, (long long)t, (long long)tile * 16, lds, v, st1, hook, hook2);
    };
    fused_list_prefetch<per0, per1>(f.c, s_item, l0, l1, t0, t1);
}

// 2-D shapes on the 512-thread tiles (axis length 512 * A): fp64 1024 x 1024 (the published double-precision shape), fp32 with a
// 2048-point axis -- fft_fused2d_kernel's data flow
template <typename T, int A0, int A1, bool SPLIT, bool NT>
__global__ void __launch_bounds__(512, 2) fft_fused3d_kernel(const FusedArgs f) {
    __shared__ __attribute__((aligned(16))) T lds[Col3Lds<T, true>::SCALARS];
    __shared__ unsigned s_item;
    constexpr unsigned per0 = A0 > A1 ? 1u : (unsigned)(A1 / A0);
    constexpr unsigned per1 = A1 > A0 ? 1u : (unsigned)(A0 / A1);
    fused_loop<per0, per1, false>(
        f.c, &s_item,
        [&](unsigned t, unsigned slot, unsigned tile, auto hook) {
            col3_tile<T, A0, true, false, SPLIT, NT, false, false, true>(f.p0, (long long)t, (long long)slot, (long long)tile * 16, lds, hook);
        },
        [&](unsigned slot, unsigned t, unsigned tile, auto hook) {
            col3_tile<T, A1, true, false, false, false, NT, SPLIT>(f.p1, (long long)slot, (long long)t, (long long)tile * 16, lds, hook);
        });
}

}  // namespace mifft

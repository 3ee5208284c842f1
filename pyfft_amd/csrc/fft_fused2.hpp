// Fused two-pass kernel for a long contiguous axis N = L0 * L1 (L0, L1 in {256, 512, 1024}): both Stockham
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

struct FusedArgs {
    TileArgs p0;         // pass 0: in = user input,  out = scratch ring (matrix index = ring slot)
    TileArgs p1;         // pass 1: in = scratch ring, out = user output
    unsigned* counters;  // [0] ticket, [1] error, [2 .. 2+batch) wdone, [2+batch .. 2+2*batch) rdone  (zeroed per launch)
    unsigned batch;      // number of transforms
    unsigned lag;        // pass 1 of transform t is queued with pass 0 of transform t + lag
    unsigned ring;       // scratch ring slots (transforms); ring > lag
    unsigned tiles0;     // 16-column tiles per transform in pass 0 (= L1 / 16)
    unsigned tiles1;     // 16-column tiles per transform in pass 1 (= L0 / 16)
};

// Deferred publish of a pass-0 tile: its write-through stores drain under the first poll of the NEXT item's dependency, and
// the counter is bumped BEFORE this work-group starts to wait.  (Publishing only after the wait has ended deadlocks: work-group
// X owes a tile of transform T and waits for U while Y owes a tile of U and waits for T -- measured as dependency time-outs.)
struct FusedPending {
    unsigned* ctr;   // wdone counter still to be bumped, or nullptr
};

// wait until *ctr >= target (one lane polls, bounded); ACQ: also make other work-groups' published data visible
template <bool ACQ> __device__ __forceinline__ void fused_wait_ge(unsigned* ctr, unsigned target, unsigned* err, FusedPending& pend) {
    unsigned seen = 0;
    if (threadIdx.x == 0) seen = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // first poll, in flight
    if (pend.ctr != nullptr) {   // (uniform) the owed tile: every wave drains its stores, the barrier joins them, lane 0 publishes
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(pend.ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pend.ctr = nullptr;
    }
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        while (seen < target) {
            __builtin_amdgcn_s_sleep(32);
            if (++spins > (1u << 22)) {  // ~ seconds: never hang the GPU
                __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            seen = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

template <typename T, int A0, int A1, bool SPLIT, bool NT>
__global__ void __launch_bounds__(256, 2) fft_fused2_kernel(const FusedArgs f) {
    constexpr int E0 = Col2Lds<A0, true>::ELEMS, E1 = Col2Lds<A1, false>::ELEMS;
    __shared__ __attribute__((aligned(16))) cplx<T> lds[E0 > E1 ? E0 : E1];
    __shared__ unsigned s_item;

    unsigned* const next = f.counters;
    unsigned* const err = f.counters + 1;
    unsigned* const wdone = f.counters + 2;
    unsigned* const rdone = wdone + f.batch;
    // a group = the tiles0 pass-0 tiles of transform g and the tiles1 pass-1 tiles of transform g - lag, interleaved in their
    // ratio (tiles0 : tiles1 = L1 : L0 = A1 : A0), so that no ticket is an empty item
    constexpr unsigned per0 = A0 > A1 ? 1u : (unsigned)(A1 / A0);
    constexpr unsigned per1 = A1 > A0 ? 1u : (unsigned)(A0 / A1);
    constexpr unsigned period = per0 + per1;
    const unsigned gsize = f.tiles0 + f.tiles1;
    const unsigned total = (f.batch + f.lag) * gsize;

    // the ticket of the NEXT item is drawn while the current one is being worked on (the returning atomic takes 1-3 us under
    // load: MI355X_MICROARCH.md, dequeue row), so its latency is off the critical path
    FusedPending pend = {nullptr};
    unsigned ahead = 0;
    if (threadIdx.x == 0) ahead = __hip_atomic_fetch_add(next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
        __syncthreads();  // the previous item's LDS traffic and its s_item read are over
        if (threadIdx.x == 0) {
            s_item = ahead;
            if (ahead < total) ahead = __hip_atomic_fetch_add(next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        const unsigned item = s_item;
        if (item >= total) break;
        const unsigned g = item / gsize, k = item % gsize, j = k / period, m = k % period;
        if (m < per0) {
            const unsigned tile = j * per0 + m;
            if (g >= f.batch) {          // (the drain of the last `lag` transforms) nothing to do, but never sit on a publish
                fused_flush(pend);
                continue;
            }
            const unsigned t = g;
            if (t >= f.ring) {
                fused_wait_ge<false>(rdone + (t - f.ring), f.tiles1, err, pend);
            } else {
                fused_flush(pend);
            }
            col2_tile<T, A0, true, true, SPLIT, true, NT, false, false>(f.p0, (long long)t, (long long)(t % f.ring), (long long)tile * 16, lds);
            pend.ctr = wdone + t;       // published behind the next item's dependency wait (or at the end)
        } else {
            const unsigned tile = j * per1 + (m - per0);
            if (g < f.lag) {            // (the fill of the first `lag` transforms)
                fused_flush(pend);
                continue;
            }
            const unsigned t = g - f.lag;
            // (reading the ring with sc1 loads instead of the acquire fence measured the same -- 22.52 ms -- and 8-byte sc1 loads
            // at two work-groups per CU are outside the hand-off forms MI355X_MICROARCH.md lists as validated: the fence stays)
            fused_wait_ge<true>(wdone + t, f.tiles0, err, pend);
            col2_tile<T, A1, false, false, false, false, false, NT, SPLIT>(f.p1, (long long)(t % f.ring), (long long)t, (long long)tile * 16, lds);
            fused_signal_read(rdone + t);
        }
    }
    fused_flush(pend);
}

// The same work list with the 512-thread tiles of fft_col3.hpp (L = 512 * A): fp32 N = 2^22 = 2048 x 2048 (BASELINE config 5;
// one work-group per CU, the ring holds 7 transforms of 32 MiB), fp32 N = 2^21 = 2048 x 1024 and fp64 N = 2^20 = 1024 x 1024
// (14 transforms of 16 MiB).  A group is tiles0 + tiles1 items with the two kinds interleaved in their ratio (1:1 or 1:2).
template <typename T, int A0, int A1, bool SPLIT, bool NT>
__global__ void __launch_bounds__(512, 2) fft_fused3_kernel(const FusedArgs f) {
    constexpr int E0 = Col3Lds<T, true>::SCALARS, E1 = Col3Lds<T, false>::SCALARS;
    __shared__ __attribute__((aligned(16))) T lds[E0 > E1 ? E0 : E1];
    __shared__ unsigned s_item;

    unsigned* const next = f.counters;
    unsigned* const err = f.counters + 1;
    unsigned* const wdone = f.counters + 2;
    unsigned* const rdone = wdone + f.batch;
    constexpr unsigned per0 = A0 > A1 ? 1u : (unsigned)(A1 / A0);   // tiles0 : tiles1 = L1 : L0 = A1 : A0
    constexpr unsigned per1 = A1 > A0 ? 1u : (unsigned)(A0 / A1);
    constexpr unsigned period = per0 + per1;
    const unsigned gsize = f.tiles0 + f.tiles1;
    const unsigned total = (f.batch + f.lag) * gsize;

    FusedPending pend = {nullptr};
    unsigned ahead = 0;
    if (threadIdx.x == 0) ahead = __hip_atomic_fetch_add(next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) {
            s_item = ahead;
            if (ahead < total) ahead = __hip_atomic_fetch_add(next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        const unsigned item = s_item;
        if (item >= total) break;
        const unsigned g = item / gsize, k = item % gsize, j = k / period, m = k % period;
        if (m < per0) {
            const unsigned tile = j * per0 + m;
            if (g >= f.batch) {
                fused_flush(pend);
                continue;
            }
            const unsigned t = g;
            if (t >= f.ring) fused_wait_ge<false>(rdone + (t - f.ring), f.tiles1, err, pend);
            else fused_flush(pend);
            col3_tile<T, A0, true, true, SPLIT, NT, false, false, true>(f.p0, (long long)t, (long long)(t % f.ring), (long long)tile * 16, lds);
            pend.ctr = wdone + t;
        } else {
            const unsigned tile = j * per1 + (m - per0);
            if (g < f.lag) {
                fused_flush(pend);
                continue;
            }
            const unsigned t = g - f.lag;
            fused_wait_ge<true>(wdone + t, f.tiles0, err, pend);
            col3_tile<T, A1, false, false, false, false, NT, SPLIT>(f.p1, (long long)(t % f.ring), (long long)t, (long long)tile * 16, lds);
            fused_signal_read(rdone + t);
        }
    }
    fused_flush(pend);
}

}  // namespace mifft

// Persistent form of a 3-D transform made of two PASS PAIRS (fft_pair.hpp): the (ROW x, COL y R0) tiles of every transform are the
// pass-0 items and its (COL y R1, COL z) tiles the pass-1 items of the dependency-ordered work list of fft_fused2.hpp, so that the
// buffer between the two pairs is a ring of a few whole transforms that lives in the Infinity Cache and there is no launch
// boundary -- what took 1024 x 1024 from 0.355 (two launches per cache-sized chunk) to 0.445.  For the cubes whose transform is a
// fraction of that cache: 128^3 (16 MiB fp32 / 32 MiB fp64), the published row of the reference (doc/source/index.rst:373,
// test/test_performance.py:40-44; chain of passes per axis pyfft/plan.py:160-167).
//   pass-0 item (t, tile): tile = plane * R1 + l   -- rows y = r * R1 + l of plane `plane` of transform t -> ring slot, write-through
//   pass-1 item (t, tile): tile = group of W adjacent elements of [R0][nx] -- all (r1, z) of them from the ring slot -> transform t
//   dependency: a pass-1 tile reads one segment of EVERY plane, so it waits for all pass-0 tiles of its transform (wdone[t] == tiles0);
//   a pass-0 tile of t waits until the slot's previous owner t - ring has been read by all its pass-1 tiles (rdone).
// The work-group is persistent, so both tile kinds run on the YZ tile's thread count (see SUB0 below); the LDS array is the larger need.
#pragma once
#include "fft_fused2.hpp"
#include "fft_pair.hpp"

namespace mifft {

struct FusedPairArgs {
    PairArgs a0;       // XY pair: in0 (+ in1: split-complex planes) = user input, out0 = ring (always interleaved)
    PairArgs a1;       // YZ pair: in0 = ring, out0 (+ out1) = user output
    FusedCtl c;
    long long n;       // elements per transform (= ring slot pitch)
};

// The persistent work-group has C1::NT threads.  An XY tile kind with fewer threads (fp32 128^3: 256 against the YZ tile's 512) runs
// SUB0 = C1::NT / C0::NT tiles side by side, each on its own slice of the thread index and of the LDS array, in lock step (same
// tile code, so the same barriers): a pass-0 ITEM is then SUB0 consecutive tiles.  (First form of round 4: both kinds on 256
// threads, the YZ tile at 32 points per thread -- 8 waves per CU could not cover the latency of 32 KiB tiles: 0.265 against 0.351
// for the pipelined chunks, profiles/r04_b_cube_sweep.log.)
template <typename T, typename C0, typename C1, unsigned PER0, unsigned PER1>
__global__ void __launch_bounds__(C1::NT, C1::NT >= 512 ? 4 : 2) fft_fusedp_kernel(const FusedPairArgs f) {
    static_assert(C1::NT % C0::NT == 0, "the XY tile kind runs on a whole fraction of the work-group");
    static_assert(C0::HALF == C1::HALF && C0::SPLIT_IN == C1::SPLIT_OUT, "the same exchange form; split planes on both user sides or on neither");
    constexpr int SUB0 = C1::NT / C0::NT;
    constexpr int LDS0 = C0::P + C0::P / 16, LDS1 = C1::P + C1::P / 16;
    constexpr int LDSN = SUB0 * LDS0 > LDS1 ? SUB0 * LDS0 : LDS1;
    using LdsT = typename std::conditional<C0::HALF, T, cplx<T>>::type;
    __shared__ __attribute__((aligned(16))) LdsT lds[LDSN];
    __shared__ unsigned s_item;
    using M0 = typename C0::MAP;
    using M1 = typename C1::MAP;
    fused_loop<PER0, PER1, false>(
        f.c, &s_item,
        [&](unsigned t, unsigned slot, unsigned item, auto) {
            // (an opaque copy of the thread index: what a tile derives from it is recomputed per tile instead of being hoisted out
            // of the persistent loop for both tile kinds at once)
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));
            // (wave-uniform: C0::NT is a multiple of the wave size; said so explicitly, the tile's base addresses stay in SGPRs)
            const unsigned sub = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)tid / (unsigned)C0::NT));
            const unsigned tile = item * (unsigned)SUB0 + sub;
            const unsigned l = tile % (unsigned)M0::C0;
            const long long plane = tile / (unsigned)M0::C0;
            const long long rel_in = plane * M0::BOUTER + (long long)l * M0::BI0, rel_out = plane * M0::BOUTER + (long long)l * M0::BO0;
            pair_tile<T, C0>(f.a0, (long long)t * f.n + rel_in, (long long)slot * f.n + rel_out, (int)l, lds + sub * LDS0, tid % C0::NT, true, 2);
        },
        [&](unsigned slot, unsigned t, unsigned tile, auto) {
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));
            pair_tile<T, C1>(f.a1, (long long)slot * f.n + (long long)tile * M1::BI0, (long long)t * f.n + (long long)tile * M1::BO0, (int)tile,
                             lds, tid, false, 1);
        });
}

template <typename T, typename C0, typename C1, unsigned PER0, unsigned PER1>
static inline int launch_fusedp(const FusedPairArgs* f, unsigned grid, hipStream_t s) {
    // a 1024-thread work-group fills a CU: never more persistent work-groups than CUs
    if (C1::NT >= 1024) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0 &&
            grid > (unsigned)cus)
            grid = (unsigned)cus;
    }
    hipLaunchKernelGGL((fft_fusedp_kernel<T, C0, C1, PER0, PER1>), dim3(grid), dim3(C1::NT), 0, s, *f);
    return (int)hipGetLastError();
}

}  // namespace mifft

// Instances of the persistent two-pair kernel (fft_fusedp.hpp): the same tile configurations as the plain pair launches of
// fft_pair_f32.hip / fft_pair_f64.hip (same stage lists = the same arithmetic, bit for bit), with both tile kinds on 256 threads.
#include "mifft_internal.h"
#include "fft_fusedp.hpp"

using namespace mifft;

// *tiles0 / *tiles1 = tiles per transform of the two pairs; query != 0: nothing is launched
extern "C" int mifft_fusedp(int f64, int split, int x, int y, int z, const FusedPairArgs* f, unsigned grid, hipStream_t s, int query, int* r0,
                            unsigned* tiles0, unsigned* tiles1) {
    if (split) return mifft_fusedp_more(f64, split, x, y, z, f, grid, s, query, r0, tiles0, tiles1);   // split planes: fft_fusedp2.hip
#define RL(...) RadixList<__VA_ARGS__>
    // development switch (A/B, MIFFT_PAIR): 3 = the first form of this kernel (narrow fp64 tiles, the (16, 2) y list in fp32)
    const int variant = mifft_debug_get(MIFFT_DEBUG_PAIR) == 3 ? 3 : 0;
#define CASE(T, F64, NX, NY, NZ, R0, R1, W, XY, YZ, VARIANT)                                      \
    if (f64 == F64 && x == NX && y == NY && z == NZ && variant == VARIANT) {                     \
        constexpr unsigned t0 = (unsigned)NZ * R1 / (YZ::NT / XY::NT), t1 = (unsigned)NX * R0 / W; \
        constexpr unsigned per0 = t0 >= t1 ? t0 / t1 : 1u, per1 = t1 > t0 ? t1 / t0 : 1u;       \
        static_assert(t0 * per1 == t1 * per0, "item counts must be in a small integer ratio");    \
        if (r0) *r0 = R0;                                                                         \
        if (tiles0) *tiles0 = t0;                                                                 \
        if (tiles1) *tiles1 = t1;                                                                 \
        return query ? 0 : launch_fusedp<T, XY, YZ, per0, per1>(f, grid, s);                     \
    }
    // ---- 128^3 fp64, y = 32 x 4: two XY tiles (128 x 32 = 4096 points on 256 threads each) side by side, YZ tile 16 x 4 x 128 = 8192
    // points on 512 threads (256-byte segments); half-form exchanges, 69.6 KiB of LDS, two work-groups = 16 waves per CU; every item
    // moves 128 KiB in and 128 KiB out, like a tile of the 1-D kernel.  256 + 256 items per transform.  The stage lists are those of
    // fft_pair_f64.hip, so the arithmetic is the plain pair launches'.  0.442 of the roofline at 4 GiB, 0.412 at 1 GiB (pipelined
    // chunks: 0.357 / 0.338; first form with 64 KiB tiles and 128-byte segments: 0.386 / 0.373 -- profiles/r04_f_cube_wide_tiles.log)
    using XY128d = PairXY<double, 128, 32, 4, 256, true, 1, RL(8, 16), RL(8, 4), false>;
    using YZ128d = PairYZ<double, 128 * 32, 4, 128, 16, 512, true, 1, RL(4), RL(8, 16), false>;
    CASE(double, 1, 128, 128, 128, 32, 4, 16, XY128d, YZ128d, 0)
    using XY128dn = PairXY<double, 128, 32, 4, 256, false, 1, RL(8, 16), RL(8, 4), false>;
    using YZ128dn = PairYZ<double, 128 * 32, 4, 128, 8, 256, false, 1, RL(4), RL(8, 16), false>;
    CASE(double, 1, 128, 128, 128, 32, 4, 8, XY128dn, YZ128dn, 3)
    // ---- 128^3 fp32: the XY tile (128 x 32 = 4096 points) on 128 threads x 32 points with the y digit as ONE radix-32 stage -- four LDS
    // exchanges per point instead of the five of the (16, 2) list -- and FOUR of them side by side: a pass-0 item is a whole 128 KiB
    // plane; YZ tile 16 x 4 x 128 = 8192 points on 512 threads; half-form exchanges, 69.6 KiB of LDS, two work-groups per CU.
    // 0.357 (pipelined chunks) -> 0.423 at 4 GiB (profiles/r04_l_anchored_twiddles_fp64_mid.log).  The fp32 cube is bound by work per
    // POINT, not by bytes (11.5 us per 2^21 points against 9.4 us for the 1-D kernel with two exchanges), so what moved it was
    // fewer exchanges (this list: + 3 points) and cheaper inter-pass twiddles (fft_pair.hpp pair_store: + 3 points); WIDER tiles did
    // not: 32-column YZ tiles at 32 points per thread 0.354 (16 spilled registers), 1024-thread work-groups 0.341 against 0.366 for
    // the list's first form (profiles/r04_g_cube_fp32_forms.log; those two instances are no longer built).
    using XY128f = PairXY<float, 128, 32, 4, 128, true, 1, RL(8, 16), RL(32), false>;
    using YZ128f = PairYZ<float, 128 * 32, 4, 128, 16, 512, true, 1, RL(4), RL(8, 16), false>;
    CASE(float, 0, 128, 128, 128, 32, 4, 16, XY128f, YZ128f, 0)
    // (MIFFT_PAIR=3: the first form -- two 256-thread XY tiles with the (16, 2) list per item, full-complex exchanges)
    using XY128fn = PairXY<float, 128, 32, 4, 256, false, 1, RL(8, 16), RL(16, 2), false>;
    using YZ128fn = PairYZ<float, 128 * 32, 4, 128, 16, 512, false, 1, RL(4), RL(8, 16), false>;
    CASE(float, 0, 128, 128, 128, 32, 4, 16, XY128fn, YZ128fn, 3)
#undef CASE
#undef RL
    return variant == 0 ? mifft_fusedp_more(f64, 0, x, y, z, f, grid, s, query, r0, tiles0, tiles1) : -2;   // fft_fusedp2.hip
}

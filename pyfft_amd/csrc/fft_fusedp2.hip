// More instances of the persistent two-pair kernel (fft_fusedp.hpp): 3-D shapes with 64- and 128-point axes next to the 128^3 cubes
// of fft_fusedp.hip (second batch of round 4).  These shapes run as ONE plane tile (fft_nd2.hpp) + one strided z pass per
// cache-sized chunk otherwise (0.33-0.40 of the roofline, profiles/r04_t_tail_survey.log); only the persistent form factors
// their y axis R0 x R1 -- the plan builds the four-pass list for it alone (pyfft_amd/plan.py _pair_alt).
//   XY tile = NX x R0 points (strided rows in, R0 contiguous rows out); SUB0 = NT(YZ) / NT(XY) of them side by side per item
//   YZ tile = W x R1 x NZ points, W adjacent elements of [R0][NX]: W = 32 where NZ = 64 (256-byte segments, 8192-point tiles)
#include "mifft_internal.h"
#include "fft_fusedp.hpp"

using namespace mifft;

extern "C" int mifft_fusedp_split(int f64, int x, int y, int z, const FusedPairArgs* f, unsigned grid, hipStream_t s, int query, int* r0,
                                  unsigned* tiles0, unsigned* tiles1);   // fft_fusedp3.hip

extern "C" int mifft_fusedp_more(int f64, int split, int x, int y, int z, const FusedPairArgs* f, unsigned grid, hipStream_t s, int query, int* r0,
                                 unsigned* tiles0, unsigned* tiles1) {
    if (split) return mifft_fusedp_split(f64, x, y, z, f, grid, s, query, r0, tiles0, tiles1);
#define RL(...) RadixList<__VA_ARGS__>
#define CASE(T, F64, NX, NY, NZ, R0, R1, W, XY, YZ)                                               \
    if (f64 == F64 && x == NX && y == NY && z == NZ) {                                            \
        constexpr unsigned t0 = (unsigned)NZ * R1 / (YZ::NT / XY::NT), t1 = (unsigned)NX * R0 / W; \
        constexpr unsigned per0 = t0 >= t1 ? t0 / t1 : 1u, per1 = t1 > t0 ? t1 / t0 : 1u;       \
        static_assert(t0 * per1 == t1 * per0, "item counts must be in a small integer ratio");    \
        if (r0) *r0 = R0;                                                                         \
        if (tiles0) *tiles0 = t0;                                                                 \
        if (tiles1) *tiles1 = t1;                                                                 \
        return query ? 0 : launch_fusedp<T, XY, YZ, per0, per1>(f, grid, s);                     \
    }
    // Every shape factors y = R0 x 4.  XY tile (NX x R0 points; SUB0 = 512 / NT of them side by side per item), by (nx, ny):
    //   (128, 128)  R0 = 32: the cube's tiles (fp32 128 threads x 32 points, fp64 256 x 16)
    //   ( 64, 128)  R0 = 32: 2048 points (fp32 one wave x 32 points, fp64 128 threads x 16)
    //   (128,  64)  R0 = 16: 2048 points (the same thread counts)
    //   ( 64,  64)  R0 = 16: 1024 points on one wave x 16 points
    using XY128x128f = PairXY<float, 128, 32, 4, 128, true, 1, RL(8, 16), RL(32), false>;
    using XY128x128d = PairXY<double, 128, 32, 4, 256, true, 1, RL(8, 16), RL(8, 4), false>;
    using XY64x128f = PairXY<float, 64, 32, 4, 64, true, 1, RL(4, 16), RL(32), false>;
    using XY64x128d = PairXY<double, 64, 32, 4, 128, true, 1, RL(4, 16), RL(8, 4), false>;
    using XY128x64f = PairXY<float, 128, 16, 4, 64, true, 1, RL(8, 16), RL(16), false>;
    using XY128x64d = PairXY<double, 128, 16, 4, 128, true, 1, RL(8, 16), RL(16), false>;
    using XY64x64f = PairXY<float, 64, 16, 4, 64, true, 1, RL(4, 16), RL(16), false>;
    using XY64x64d = PairXY<double, 64, 16, 4, 64, true, 1, RL(4, 16), RL(16), false>;
    // round 6: (32, 32, 128), a shape of the reference's own benchmark list (test/test_performance.py:40-44) -- y = 8 x 4, the XY tile 128 x 8
    // = 1024 points on one wave x 16 points, the YZ tile 64 x 4 x 32 = 8192 points; pipelined chunks -> persistent at 1 GiB per side:
    // see the CASE lines below
    using XY128x32f = PairXY<float, 128, 8, 4, 64, true, 1, RL(8, 16), RL(8), false>;
    using XY128x32d = PairXY<double, 128, 8, 4, 64, true, 1, RL(8, 16), RL(8), false>;
    // YZ tile: W x 4 x NZ = 8192 points on 512 threads x 16 points -- W = 32 adjacent elements of [R0][NX] for nz = 64, 16 for nz = 128
    // (64 for nz = 32)
#define YZ32(T, S0) PairYZ<T, S0, 4, 32, 64, 512, true, 1, RL(4), RL(2, 16), false>
#define YZ64(T, S0) PairYZ<T, S0, 4, 64, 32, 512, true, 1, RL(4), RL(4, 16), false>
#define YZ128(T, S0) PairYZ<T, S0, 4, 128, 16, 512, true, 1, RL(4), RL(8, 16), false>
    //   (nz, ny, nx)                 NX   NY   NZ  R0 R1  W
    CASE(float, 0, 128, 128, 64, 32, 4, 32, XY128x128f, YZ64(float, 128 * 32))       // (64, 128, 128)
    CASE(double, 1, 128, 128, 64, 32, 4, 32, XY128x128d, YZ64(double, 128 * 32))
    CASE(float, 0, 64, 128, 128, 32, 4, 16, XY64x128f, YZ128(float, 64 * 32))        // (128, 128, 64)
    CASE(double, 1, 64, 128, 128, 32, 4, 16, XY64x128d, YZ128(double, 64 * 32))
    CASE(float, 0, 64, 128, 64, 32, 4, 32, XY64x128f, YZ64(float, 64 * 32))          // (64, 128, 64)
    CASE(double, 1, 64, 128, 64, 32, 4, 32, XY64x128d, YZ64(double, 64 * 32))
    CASE(float, 0, 128, 64, 128, 16, 4, 16, XY128x64f, YZ128(float, 128 * 16))       // (128, 64, 128)
    CASE(double, 1, 128, 64, 128, 16, 4, 16, XY128x64d, YZ128(double, 128 * 16))
    CASE(float, 0, 128, 64, 64, 16, 4, 32, XY128x64f, YZ64(float, 128 * 16))         // (64, 64, 128)
    CASE(double, 1, 128, 64, 64, 16, 4, 32, XY128x64d, YZ64(double, 128 * 16))
    CASE(float, 0, 64, 64, 128, 16, 4, 16, XY64x64f, YZ128(float, 64 * 16))          // (128, 64, 64)
    CASE(double, 1, 64, 64, 128, 16, 4, 16, XY64x64d, YZ128(double, 64 * 16))
    CASE(float, 0, 64, 64, 64, 16, 4, 32, XY64x64f, YZ64(float, 64 * 16))            // (64, 64, 64)
    CASE(double, 1, 64, 64, 64, 16, 4, 32, XY64x64d, YZ64(double, 64 * 16))
    CASE(float, 0, 128, 32, 32, 8, 4, 64, XY128x32f, YZ32(float, 128 * 8))             // (32, 32, 128)
    CASE(double, 1, 128, 32, 32, 8, 4, 64, XY128x32d, YZ32(double, 128 * 8))
    // ... and its neighbours with one or two 32-point axes that the existing tile kinds cover (1 GiB per side, pipelined chunks -> persistent:
    // profiles/r06_o_cube32_neighbours.log)
    CASE(float, 0, 128, 64, 32, 16, 4, 64, XY128x64f, YZ32(float, 128 * 16))           // (32, 64, 128)
    CASE(double, 1, 128, 64, 32, 16, 4, 64, XY128x64d, YZ32(double, 128 * 16))
    CASE(float, 0, 128, 128, 32, 32, 4, 64, XY128x128f, YZ32(float, 128 * 32))         // (32, 128, 128)
    CASE(double, 1, 128, 128, 32, 32, 4, 64, XY128x128d, YZ32(double, 128 * 32))
    CASE(float, 0, 128, 32, 64, 8, 4, 32, XY128x32f, YZ64(float, 128 * 8))             // (64, 32, 128)
    CASE(double, 1, 128, 32, 64, 8, 4, 32, XY128x32d, YZ64(double, 128 * 8))
    CASE(float, 0, 128, 32, 128, 8, 4, 16, XY128x32f, YZ128(float, 128 * 8))           // (128, 32, 128)
    CASE(double, 1, 128, 32, 128, 8, 4, 16, XY128x32d, YZ128(double, 128 * 8))
    CASE(float, 0, 64, 64, 32, 16, 4, 64, XY64x64f, YZ32(float, 64 * 16))              // (32, 64, 64)
    CASE(double, 1, 64, 64, 32, 16, 4, 64, XY64x64d, YZ32(double, 64 * 16))
    CASE(float, 0, 64, 128, 32, 32, 4, 64, XY64x128f, YZ32(float, 64 * 32))            // (32, 128, 64)
    CASE(double, 1, 64, 128, 32, 32, 4, 64, XY64x128d, YZ32(double, 64 * 32))
#undef YZ32
#undef YZ64
#undef YZ128
#undef CASE
#undef RL
    return -2;
}

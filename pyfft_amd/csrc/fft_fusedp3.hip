// The persistent two-pair kernel (fft_fusedp.hpp) on SPLIT-COMPLEX user buffers (re / im planes; the reference's float32 / float64
// dtypes, pyfft/plan.py:10-63): the XY tiles read the two planes, the YZ tiles write them, the ring between them is interleaved.
// Every shape of {64, 128}^3 in both precisions.  The READ side of the planes is whole x rows (256 / 512 bytes per plane), so the
// half-line over-fetch of fft_fused2.hpp's split note cannot happen; on the WRITE side a YZ segment is a whole 128-byte line per
// plane except for fp32 with nz = 128 (16 elements = 64 bytes: partial-line writes, which cost no extra traffic -- those shapes
// gain 5-9 points instead of 7-11).  Same tile geometry and stage lists as fft_fusedp.hip / fft_fusedp2.hip.
// 2 GiB per side, pipelined chunks -> persistent (profiles/r04_aa_pair_split_planes.log): fp64 0.26-0.32 -> 0.44-0.46, fp32 with
// nz = 64 0.29-0.32 -> 0.38-0.40, with nz = 128 0.29-0.32 -> 0.37-0.38.
#include "mifft_internal.h"
#include "fft_fusedp.hpp"

using namespace mifft;

extern "C" int mifft_fusedp_split(int f64, int x, int y, int z, const FusedPairArgs* f, unsigned grid, hipStream_t s, int query, int* r0,
                                  unsigned* tiles0, unsigned* tiles1) {
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
    using XY128x128f = PairXY<float, 128, 32, 4, 128, true, 1, RL(8, 16), RL(32), true>;
    using XY128x128d = PairXY<double, 128, 32, 4, 256, true, 1, RL(8, 16), RL(8, 4), true>;
    using XY64x128f = PairXY<float, 64, 32, 4, 64, true, 1, RL(4, 16), RL(32), true>;
    using XY64x128d = PairXY<double, 64, 32, 4, 128, true, 1, RL(4, 16), RL(8, 4), true>;
    using XY128x64f = PairXY<float, 128, 16, 4, 64, true, 1, RL(8, 16), RL(16), true>;
    using XY128x64d = PairXY<double, 128, 16, 4, 128, true, 1, RL(8, 16), RL(16), true>;
    using XY64x64f = PairXY<float, 64, 16, 4, 64, true, 1, RL(4, 16), RL(16), true>;
    using XY64x64d = PairXY<double, 64, 16, 4, 64, true, 1, RL(4, 16), RL(16), true>;
#define YZ64(T, S0) PairYZ<T, S0, 4, 64, 32, 512, true, 1, RL(4), RL(4, 16), true>
#define YZ128(T, S0) PairYZ<T, S0, 4, 128, 16, 512, true, 1, RL(4), RL(8, 16), true>
    //   (nz, ny, nx)                 NX   NY   NZ  R0 R1  W
    CASE(float, 0, 128, 128, 64, 32, 4, 32, XY128x128f, YZ64(float, 128 * 32))       // (64, 128, 128)
    CASE(float, 0, 64, 128, 64, 32, 4, 32, XY64x128f, YZ64(float, 64 * 32))          // (64, 128, 64)
    CASE(float, 0, 128, 64, 64, 16, 4, 32, XY128x64f, YZ64(float, 128 * 16))         // (64, 64, 128)
    CASE(float, 0, 64, 64, 64, 16, 4, 32, XY64x64f, YZ64(float, 64 * 16))            // (64, 64, 64)
    CASE(float, 0, 128, 128, 128, 32, 4, 16, XY128x128f, YZ128(float, 128 * 32))     // (128, 128, 128)  (z = 128: 64-byte segments per plane on the WRITE side)
    CASE(float, 0, 64, 128, 128, 32, 4, 16, XY64x128f, YZ128(float, 64 * 32))        // (128, 128, 64)
    CASE(float, 0, 128, 64, 128, 16, 4, 16, XY128x64f, YZ128(float, 128 * 16))       // (128, 64, 128)
    CASE(float, 0, 64, 64, 128, 16, 4, 16, XY64x64f, YZ128(float, 64 * 16))          // (128, 64, 64)
    CASE(double, 1, 128, 128, 128, 32, 4, 16, XY128x128d, YZ128(double, 128 * 32))   // (128, 128, 128)
    CASE(double, 1, 128, 128, 64, 32, 4, 32, XY128x128d, YZ64(double, 128 * 32))     // (64, 128, 128)
    CASE(double, 1, 64, 128, 128, 32, 4, 16, XY64x128d, YZ128(double, 64 * 32))      // (128, 128, 64)
    CASE(double, 1, 64, 128, 64, 32, 4, 32, XY64x128d, YZ64(double, 64 * 32))        // (64, 128, 64)
    CASE(double, 1, 128, 64, 128, 16, 4, 16, XY128x64d, YZ128(double, 128 * 16))     // (128, 64, 128)
    CASE(double, 1, 128, 64, 64, 16, 4, 32, XY128x64d, YZ64(double, 128 * 16))       // (64, 64, 128)
    CASE(double, 1, 64, 64, 128, 16, 4, 16, XY64x64d, YZ128(double, 64 * 16))        // (128, 64, 64)
    CASE(double, 1, 64, 64, 64, 16, 4, 32, XY64x64d, YZ64(double, 64 * 16))          // (64, 64, 64)
#undef YZ64
#undef YZ128
#undef CASE
#undef RL
    return -2;
}

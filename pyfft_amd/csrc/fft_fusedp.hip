// Instances of the persistent two-pair kernel (fft_fusedp.hpp): the same tile configurations as the plain pair launches of
// fft_pair_f32.hip / fft_pair_f64.hip (same stage lists = the same arithmetic, bit for bit), with both tile kinds on 256 threads.
#include "mifft_internal.h"
#include "fft_fusedp.hpp"

using namespace mifft;

// *tiles0 / *tiles1 = tiles per transform of the two pairs; query != 0: nothing is launched
extern "C" int mifft_fusedp(int f64, int x, int y, int z, const FusedPairArgs* f, unsigned grid, hipStream_t s, int query, int* r0,
                            unsigned* tiles0, unsigned* tiles1) {
#define RL(...) RadixList<__VA_ARGS__>
#define CASE(T, F64, NX, NY, NZ, R0, R1, W, XY, YZ)                                               \
    if (f64 == F64 && x == NX && y == NY && z == NZ) {                                           \
        constexpr unsigned t0 = (unsigned)NZ * R1 / (YZ::NT / XY::NT), t1 = (unsigned)NX * R0 / W; \
        constexpr unsigned per0 = t0 >= t1 ? t0 / t1 : 1u, per1 = t1 > t0 ? t1 / t0 : 1u;       \
        static_assert(t0 * per1 == t1 * per0, "item counts must be in a small integer ratio");    \
        if (r0) *r0 = R0;                                                                         \
        if (tiles0) *tiles0 = t0;                                                                 \
        if (tiles1) *tiles1 = t1;                                                                 \
        return query ? 0 : launch_fusedp<T, XY, YZ, per0, per1>(f, grid, s);                     \
    }
    // 128^3 fp32: y = 32 x 4; XY tile 128 x 32 = 4096 points on 256 threads, two of them side by side; YZ tile 16 x 4 x 128 = 8192
    // points on 512 threads (the configurations of fft_pair_f32.hip): 256 + 256 items per transform, 69.6 KiB of LDS, two
    // work-groups = 16 waves per CU
    using XY128f = PairXY<float, 128, 32, 4, 256, false, 1, RL(8, 16), RL(16, 2), false>;
    using YZ128f = PairYZ<float, 128 * 32, 4, 128, 16, 512, false, 1, RL(4), RL(8, 16), false>;
    CASE(float, 0, 128, 128, 128, 32, 4, 16, XY128f, YZ128f)
    // 128^3 fp64: XY tile 4096 points, YZ tile 8 x 4 x 128 = 4096 points (64 KiB each) on 256 threads: 512 + 512 items
    using XY128d = PairXY<double, 128, 32, 4, 256, false, 1, RL(8, 16), RL(8, 4), false>;
    using YZ128d = PairYZ<double, 128 * 32, 4, 128, 8, 256, false, 1, RL(4), RL(8, 16), false>;
    CASE(double, 1, 128, 128, 128, 32, 4, 8, XY128d, YZ128d)
#undef CASE
#undef RL
    return -2;
}

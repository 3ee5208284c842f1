// Instances of the two-work-groups-per-transform form (fft_nd2z.hpp) for the one-tile shapes of 32768 points (fp32) / 16384 points (fp64)
// that tools/gen_nd2_tables.py lists as HUGE: each half runs on the "big" tile form (16384 / 8192 points, 512 threads, half-exchange
// stages, radices <= 16 / 8, two work-groups per CU).  Measured against the one-tile-per-CU kernel (profiles/r05_nd2z_two_work_groups_ab.log,
// 1 GiB per side | the reference's 32 MiB protocol): fp32 (16, 16, 128) 0.505 -> 0.637 | 0.316 -> 0.427, (128, 256) 0.480 -> 0.655 | 0.295 ->
// 0.458, (256, 128) 0.489 -> 0.620, (512, 64) 0.478 -> 0.616 | 0.289 -> 0.460, (8, 64, 64) 0.453 -> 0.570, (32, 1024) 0.550 -> 0.569, (1024, 32)
// 0.444 -> 0.474; fp64 (64, 256) 0.621 -> 0.693, (16, 16, 64) 0.634 -> 0.684, (256, 64) 0.624 -> 0.679, (16, 32, 32) 0.593 -> 0.659, (32, 512)
// 0.555 -> 0.648, (512, 32) 0.556 -> 0.633.  Three shapes LOSE and have no instance here: fp32 32^3 (0.460 / 0.455: five stages of radix <=
// 16 against three of radix 32), fp64 (128, 128) (0.631 -> 0.533) and fp64 numpy (64, 16, 16) (0.508 -> 0.486) -- their kernels spill 20-96
// bytes per lane at the two-per-CU register budget.
#include "mifft_internal.h"
#include "fft_nd2z.hpp"

using namespace mifft;

namespace {
template <typename T, int X, int Y, int Z> int go(const TileArgs* a, hipStream_t s, int query) {
    if (query) return 0;
    constexpr bool F32 = sizeof(T) == 4;
    constexpr int MAXR = F32 ? 16 : 8;
    constexpr int HY = Z > 1 ? Y : Y / 2, HZ = Z > 1 ? Z / 2 : 1;
    using CFG = Nd2zCfg<T, X, Y, Z, 512, true, 4, typename AutoRadix<X, MAXR, true>::type, typename AutoRadix<HY, MAXR, false>::type,
                        typename AutoRadix<HZ, MAXR, false>::type>;
    static_assert(CFG::P == (F32 ? 16384 : 8192), "the halves are big tiles");
    return launch_nd2z<T, CFG>(a, s);
}
}  // namespace

// 0 = launched (query: exists), -2 = no such kernel, -1 = grid too large
extern "C" int mifft_nd2z(int f64, int x, int y, int z, const TileArgs* a, hipStream_t s, int query) {
#define SHAPE(T, X, Y, Z) \
    if (x == X && y == Y && z == Z) return go<T, X, Y, Z>(a, s, query);
    if (!f64) {
        SHAPE(float, 32, 1024, 1) SHAPE(float, 64, 512, 1) SHAPE(float, 128, 256, 1) SHAPE(float, 256, 128, 1) SHAPE(float, 1024, 32, 1)
        SHAPE(float, 64, 64, 8) SHAPE(float, 128, 16, 16)
    } else {
        SHAPE(double, 32, 512, 1) SHAPE(double, 64, 256, 1) SHAPE(double, 256, 64, 1) SHAPE(double, 512, 32, 1)
        SHAPE(double, 32, 32, 16) SHAPE(double, 64, 16, 16)
    }
#undef SHAPE
    return -2;
}

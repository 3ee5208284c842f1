// (fp64 half of fft_nd2z_f32.hip)
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
// query: 1 = is there a kernel; 2 = is there one that is preferred in LARGE launches too (beyond half the last-level cache per side): the
// 32768 / 16384-point shapes unless LARGE_ONLY says the opposite -- preferred in large launches, not in small ones (shapes without a
// one-tile kernel, where the plan chooses between this launch and its two-launch chain)
template <typename T, int X, int Y, int Z, bool LARGE_ONLY = false> int go(const TileArgs* a, hipStream_t s, int query) {
    constexpr bool F32 = sizeof(T) == 4;
    if (query) return (query == 2 && (LARGE_ONLY || X * Y * Z != (F32 ? 32768 : 16384))) ? -2 : 0;
    constexpr int MAXR = F32 ? 16 : 8;
    constexpr int HY = Z > 1 ? Y : Y / 2, HZ = Z > 1 ? Z / 2 : 1;
    // halves of 16384 (fp32) / 8192 (fp64) points: the "big" tile form, half-exchange stages; halves of half that size (the two-per-CU
    // shapes themselves split in two, for SMALL launches only, see below): full-complex exchanges through 64 KiB of LDS, 16 / 8 points per thread
    constexpr int HP = X * HY * HZ;
    constexpr bool BIGHALF = HP == (F32 ? 16384 : 8192);
    static_assert(BIGHALF || HP == (F32 ? 8192 : 4096), "no tile form for this half");
    using CFG = Nd2zCfg<T, X, Y, Z, 512, BIGHALF, 4, typename AutoRadix<X, MAXR, true>::type, typename AutoRadix<HY, MAXR, false>::type,
                        typename AutoRadix<HZ, MAXR, false>::type>;
    return launch_nd2z<T, CFG>(a, s);
}

// FOUR work-groups per transform: shapes of four two-per-CU tiles (65536 points fp32 / 32768 fp64), or one-tile fp64 shapes whose halves
// spill, as quarters of 4096 points.  ALWAYS: preferred at every buffer size (query 2), else in small launches / where nothing else exists
template <typename T, int X, int Y, int Z, bool ALWAYS = false> int go4(const TileArgs* a, hipStream_t s, int query) {
    constexpr bool F32 = sizeof(T) == 4;
    if (query) return (query == 2 && !ALWAYS) ? -2 : 0;
    constexpr int MAXR = F32 ? 16 : 8;
    constexpr int QY = Z > 1 ? Y : Y / 4, QZ = Z > 1 ? Z / 4 : 1;
    constexpr int QP = X * QY * QZ;
    constexpr bool BIGQ = QP == (F32 ? 16384 : 8192);
    static_assert(BIGQ || QP == (F32 ? 8192 : 4096), "no tile form for this quarter");
    using CFG = Nd2zCfg<T, X, Y, Z, 512, BIGQ, 4, typename AutoRadix<X, MAXR, true>::type, typename AutoRadix<QY, MAXR, false>::type,
                        typename AutoRadix<QZ, MAXR, false>::type, 4>;
    return launch_nd2z<T, CFG>(a, s);
}
}  // namespace

extern "C" int mifft_nd2z_f64(int x, int y, int z, const TileArgs* a, hipStream_t s, int query) {
#define SHAPE(T, X, Y, Z) \
    if (x == X && y == Y && z == Z) return go<T, X, Y, Z>(a, s, query);
    SHAPE(double, 32, 512, 1) SHAPE(double, 64, 256, 1) SHAPE(double, 256, 64, 1) SHAPE(double, 512, 32, 1)
    SHAPE(double, 32, 32, 16) SHAPE(double, 64, 16, 16)
    // the other 16384-point shapes with x >= 16 (no one-tile kernel: two launches in place, one launch out of place -- see fft_nd2z_f32.hip)
    // 1 GiB per side, two launches -> one: (16, 1024) 0.357 -> 0.524, (1024, 16) 0.339 -> 0.538, (8, 32, 64) 0.382 -> 0.656, (32, 8, 64) 0.388 -> 0.639,
    // (8, 64, 32) 0.387 -> 0.642, (64, 8, 32) 0.371 -> 0.691, (4, 64, 64) 0.392 -> 0.683, (64, 4, 64) 0.360 -> 0.690, (32, 16, 32) 0.380 -> 0.602, (16, 64, 16)
    // 0.382 -> 0.535, (32, 32, 16) 0.383 -> 0.470; the last two lose 1-4 points at 32 MiB: LARGE_ONLY
#define SHAPEL(T, X, Y, Z) \
    if (x == X && y == Y && z == Z) return go<T, X, Y, Z, true>(a, s, query);
    SHAPE(double, 1024, 16, 1) SHAPE(double, 16, 1024, 1) SHAPE(double, 64, 32, 8) SHAPE(double, 64, 8, 32) SHAPE(double, 32, 64, 8) SHAPE(double, 32, 8, 64)
    SHAPEL(double, 16, 64, 16) SHAPE(double, 32, 16, 32) SHAPEL(double, 16, 32, 32) SHAPE(double, 64, 64, 4) SHAPE(double, 64, 4, 64)
#undef SHAPEL
    // (two-per-CU shapes, small launches only: see fft_nd2z_f32.hip)
    SHAPE(double, 16, 512, 1) SHAPE(double, 32, 256, 1) SHAPE(double, 64, 128, 1) SHAPE(double, 128, 64, 1) SHAPE(double, 256, 32, 1)
    SHAPE(double, 512, 16, 1) SHAPE(double, 32, 16, 16) SHAPE(double, 16, 32, 16) SHAPE(double, 16, 16, 32)
    // (four work-groups per transform for 32768-point shapes: the fp64 kernels spill 370-550 bytes per lane at the two-per-CU register
    // budget -- four 16-byte operands per kept point in flight, 16 points per thread -- and are not instantiated)
    // (fp64 (16, 16, 128), 32768 points, a shape of the reference's own benchmark list: four quarters of 8192 points on 1024 threads x 8 points
    // -- the 512-thread form spills 488 bytes per lane -- measured 0.347 against 0.401 for its two launches at 256 MiB and 0.369 / 0.369 at 1 GiB:
    // not instantiated)
    // the two one-tile fp64 shapes whose HALVES spill (84-96 bytes per lane), as FOUR quarters of 4096 points (8 points per thread, no
    // spills; profiles/r05_nd2z_fp64_quarters_ab.log, 32 MiB | 256 MiB | 1 GiB per side): numpy (64, 16, 16) 0.373 -> 0.502 | 0.466 -> 0.505 |
    // 0.509 -> 0.561: always; (128, 128) 0.462 -> 0.514 | 0.524 -> 0.545 | 0.606 -> 0.578: small launches only
    if (x == 16 && y == 16 && z == 64) return go4<double, 16, 16, 64, true>(a, s, query);
    if (x == 128 && y == 128 && z == 1) return go4<double, 128, 128, 1, false>(a, s, query);
#undef SHAPE
    return -2;
}

// Instances of the two-work-groups-per-transform form (fft_nd2z.hpp) for the one-tile shapes of 32768 points (fp32) / 16384 points (fp64)
// that tools/gen_nd2_tables.py lists as HUGE: each half runs on the "big" tile form (16384 / 8192 points, 512 threads, half-exchange
// stages, radices <= 16 / 8, two work-groups per CU).  Measured against the one-tile-per-CU kernel (profiles/r05_nd2z_two_work_groups_ab.log,
// 1 GiB per side | the reference's 32 MiB protocol): fp32 (16, 16, 128) 0.505 -> 0.637 | 0.316 -> 0.427, (128, 256) 0.480 -> 0.655 | 0.295 ->
// 0.458, (256, 128) 0.489 -> 0.620, (512, 64) 0.478 -> 0.616 | 0.289 -> 0.460, (8, 64, 64) 0.453 -> 0.570, (32, 1024) 0.550 -> 0.569, (1024, 32)
// 0.444 -> 0.474; fp64 (64, 256) 0.621 -> 0.693, (16, 16, 64) 0.634 -> 0.684, (256, 64) 0.624 -> 0.679, (16, 32, 32) 0.593 -> 0.659, (32, 512)
// 0.555 -> 0.648, (512, 32) 0.556 -> 0.633.  Three shapes LOSE and have no instance here: fp32 32^3 (0.460 / 0.455: five stages of radix <=
// 16 against three of radix 32), fp64 (128, 128) (0.631 -> 0.533) and fp64 numpy (64, 16, 16) (0.508 -> 0.486) -- their kernels spill 20-96
// bytes per lane at the two-per-CU register budget.
#include "../../include/mifft.h"
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

// explicit stage lists (where the automatic radix <= 16 lists lose): RS work-groups per transform on the big tile form
template <typename T, int X, int Y, int Z, typename RLX, typename RLY, typename RLZ, int RS, bool ALWAYS> int gox(const TileArgs* a, hipStream_t s, int query) {
    if (query) return (query == 2 && !ALWAYS) ? -2 : 0;
    return launch_nd2z<T, Nd2zCfg<T, X, Y, Z, 512, true, 4, RLX, RLY, RLZ, RS>>(a, s);
}

extern "C" int mifft_nd2z_f64(int x, int y, int z, const TileArgs* a, hipStream_t s, int query);   // fft_nd2z_f64.hip

// 0 = launched (query: exists), -2 = no such kernel, -1 = grid too large
extern "C" int mifft_nd2z(int f64, int x, int y, int z, const TileArgs* a, hipStream_t s, int query) {
    if (f64) return mifft_nd2z_f64(x, y, z, a, s, query);
#define SHAPE(T, X, Y, Z) \
    if (x == X && y == Y && z == Z) return go<T, X, Y, Z>(a, s, query);
#define SHAPE4(T, X, Y, Z) \
    if (x == X && y == Y && z == Z) return go4<T, X, Y, Z>(a, s, query);
    // (the one-tile shapes as FOUR quarters of 8192 points in small launches: + 1-6 points at 32 MiB, - 4-10 at 128 MiB against two halves --
    // profiles/r05_nd2z_small_launch_quarters_ab.log -- not adopted)
    SHAPE(float, 64, 512, 1) SHAPE(float, 128, 256, 1) SHAPE(float, 256, 128, 1) SHAPE(float, 64, 64, 8) SHAPE(float, 128, 16, 16)
    // the other 32768-point shapes with axes of 16 points and up: they have NO one-tile kernel (tools/gen_nd2_tables.py lists eight) and ran
    // two launches; out of place they are one launch now (the plan keeps its chain for in-place executes: MIFFT_VARIANT_OUT_OF_PLACE_ONLY).
    // 1 GiB per side, two launches -> one (profiles/r05_nd2z_shapes_without_one_tile_kernel_ab.log): (16, 2048) 0.366 -> 0.583, (2048, 16) 0.343 ->
    // 0.520, (64, 8, 64) 0.395 -> 0.538, (16, 128, 16) 0.358 -> 0.550, (128, 16, 16) 0.367 -> 0.542, (32, 16, 64) 0.350 -> 0.601, (32, 64, 16)
    // 0.339 -> 0.524, (16, 64, 32) 0.388 -> 0.515, (16, 32, 64) 0.376 -> 0.485, (64, 16, 32) 0.388 -> 0.466, (64, 512) 0.379 -> 0.443; (64, 32, 16)
    // lost (0.369 -> 0.337: its kernel spills 76 bytes per lane) and has no instance.  At 32 MiB five of them lose 1-6 points: LARGE_ONLY
#define SHAPEL(T, X, Y, Z) \
    if (x == X && y == Y && z == Z) return go<T, X, Y, Z, true>(a, s, query);
    SHAPEL(float, 512, 64, 1) SHAPE(float, 2048, 16, 1) SHAPE(float, 16, 2048, 1) SHAPE(float, 64, 8, 64) SHAPEL(float, 16, 128, 16) SHAPE(float, 16, 16, 128)
    SHAPEL(float, 64, 32, 16) SHAPE(float, 64, 16, 32) SHAPEL(float, 32, 64, 16) SHAPEL(float, 32, 16, 64) SHAPE(float, 16, 64, 32)
#undef SHAPEL
    // a 32-point axis as ONE radix-32 stage (three stages instead of five; profiles/r05_nd2z_radix32_lists_ab.log): numpy (1024, 32) 0.466 ->
    // 0.536 at 1 GiB, 0.369 -> 0.417 at 32 MiB; (32, 1024) 0.566 -> 0.657, 0.404 -> 0.452.  (Radix-32 stages next to a small remainder --
    // 64 = 2 x 32, 128 = 4 x 32 -- LOSE 3-7 points against 16 x 4 / 8 x 16 on the other shapes: those keep the automatic lists.)
    if (x == 32 && y == 1024 && z == 1) return gox<float, 32, 1024, 1, RadixList<32>, RadixList<16, 32>, RadixList<>, 2, true>(a, s, query);
    if (x == 1024 && y == 32 && z == 1) return gox<float, 1024, 32, 1, RadixList<32, 32>, RadixList<16>, RadixList<>, 2, true>(a, s, query);
    // 32^3: the x and y axes as ONE radix-32 stage each (three stages like the one-tile kernel; the automatic radix <= 16 lists take five
    // and measured 0.460 / 0.455): 0.467 -> 0.555 at 1 GiB, 0.434 -> 0.472 at 256 MiB, 0.323 -> 0.386 at 32 MiB
    // (profiles/r05_nd2z_cube32_radix32_and_eight_way_ab.log).  EIGHT work-groups per transform for (32, 32, 128) -- 131072 points, a shape
    // of the reference's own benchmark list, two launches otherwise -- measured in the same log and LOST at every size (0.406 -> 0.305 at
    // 256 MiB, 0.373 -> 0.333 at 1 GiB: eight loads per kept point, the L2 and the address path become the bound): not instantiated.
    if (x == 32 && y == 32 && z == 32) return gox<float, 32, 32, 32, RadixList<32>, RadixList<32>, RadixList<16>, 2, true>(a, s, query);
    // The two-per-CU shapes (16384 points fp32 / 8192 fp64) split in two as well -- for launches of up to half the last-level cache per
    // side (the planner's write-through rule), where a launch is a single wave of tiles and twice as many, half as long, finish sooner:
    // at the reference's 32 MiB protocol (128, 128) x 256 0.421 -> 0.573, (256, 64) 0.423 -> 0.592, (16, 32, 32) 0.393 -> 0.511, fp64
    // (128, 64) 0.535 -> 0.720, (16, 16, 32) 0.509 -> 0.670; neutral at 256 MiB, 3-7 points SLOWER at 1 GiB (fp32 (128, 128) 0.706 ->
    // 0.656), where the one-tile kernel stays (profiles/r05_nd2z_two_per_cu_shapes_ab.log).  2-D shapes with both axes >= 16 and the
    // 3-D shapes of 16 / 32-point axes.
    SHAPE(float, 16, 1024, 1) SHAPE(float, 32, 512, 1) SHAPE(float, 64, 256, 1) SHAPE(float, 128, 128, 1) SHAPE(float, 256, 64, 1)
    SHAPE(float, 512, 32, 1) SHAPE(float, 1024, 16, 1) SHAPE(float, 32, 32, 16) SHAPE(float, 32, 16, 32) SHAPE(float, 16, 32, 32)
    // four work-groups per transform: 65536-point shapes, which otherwise take two launches (the plan runs them out of place, beyond
    // half the last-level cache per side).  2 GiB per side, chain / persistent launch -> one launch (profiles/r05_nd2z_four_work_groups_ab.log):
    // (256, 256) 0.464 (persistent) -> 0.501, at 256 MiB 0.339 -> 0.444; (512, 128) 0.376 -> 0.526, (1024, 64) 0.372 -> 0.502, (64, 1024)
    // 0.380 -> 0.473, (16, 64, 64) 0.390 -> 0.468.  At 32 MiB the two launches win (0.404 against 0.356), and four shapes gain nothing or
    // lose -- (128, 512) 0.379 / 0.384 (its kernel spills), (64, 32, 32) 0.389 -> 0.358, (32, 32, 64), (16, 32, 128) -- and have no instance.
    SHAPE4(float, 256, 256, 1) SHAPE4(float, 128, 512, 1) SHAPE4(float, 64, 1024, 1) SHAPE4(float, 1024, 64, 1) SHAPE4(float, 64, 64, 16)
#undef SHAPE
#undef SHAPE4
    return -2;
}

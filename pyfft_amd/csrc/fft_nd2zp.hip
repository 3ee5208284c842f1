// Instances of the several-work-groups-per-transform kernel on dense split-complex planes (fft_nd2zp.hpp): the 32768-point float32
// shapes that run one 256 KiB tile per CU on fft_nd2p.hpp / the tiled kernel -- first of all (16, 16, 128), a shape of the reference's own
// benchmark list (test/test_performance.py:40-44) -- as two halves on the two-per-CU tile form, out of place.  Same configurations as the
// interleaved instances of fft_nd2z_f32.hip (go<float, X, Y, Z>).
#include "../../include/mifft.h"
#include "mifft_internal.h"
#include "fft_nd2zp.hpp"

using namespace mifft;

namespace {
// query: 1 = is there a kernel; 2 = is there one that is the better choice at EVERY buffer size (else: beyond half the last-level cache per
// side only -- LARGE_ONLY: the plan's two launches win in small launches)
template <typename T, int X, int Y, int Z, bool LARGE_ONLY = false> int go(const TileArgs* a, hipStream_t s, int query) {
    constexpr bool F32 = sizeof(T) == 4;
    if (query) return (query == 2 && LARGE_ONLY) ? -2 : 0;
    constexpr int MAXR = F32 ? 16 : 8;
    constexpr int HY = Z > 1 ? Y : Y / 2, HZ = Z > 1 ? Z / 2 : 1;
    constexpr int HP = X * HY * HZ;
    static_assert(HP == (F32 ? 16384 : 8192), "halves on the big tile form");
    using CFG = Nd2zCfg<T, X, Y, Z, 512, true, 4, typename AutoRadix<X, MAXR, true>::type, typename AutoRadix<HY, MAXR, false>::type,
                        typename AutoRadix<HZ, MAXR, false>::type>;
    return launch_nd2zp<T, CFG>(a, s);
}

// FOUR work-groups per transform: the 65536-point shapes (quarters of 16384 points), which take two launches otherwise; like their
// interleaved twins in large launches only (beyond half the last-level cache per side: query 2 says no)
template <typename T, int X, int Y, int Z> int go4(const TileArgs* a, hipStream_t s, int query) {
    constexpr bool F32 = sizeof(T) == 4;
    if (query) return query == 2 ? -2 : 0;
    constexpr int MAXR = F32 ? 16 : 8;
    constexpr int QY = Z > 1 ? Y : Y / 4, QZ = Z > 1 ? Z / 4 : 1;
    static_assert(X * QY * QZ == (F32 ? 16384 : 8192), "quarters on the big tile form");
    using CFG = Nd2zCfg<T, X, Y, Z, 512, true, 4, typename AutoRadix<X, MAXR, true>::type, typename AutoRadix<QY, MAXR, false>::type,
                        typename AutoRadix<QZ, MAXR, false>::type, 4>;
    return launch_nd2zp<T, CFG>(a, s);
}
}  // namespace

// 0 = launched (query: exists), -2 = no such kernel, -1 = grid too large
extern "C" int mifft_nd2zp(int f64, int x, int y, int z, const TileArgs* a, hipStream_t s, int query) {
#define SHAPE(T, F64, X, Y, Z) \
    if (f64 == F64 && x == X && y == Y && z == Z) return go<T, X, Y, Z>(a, s, query);
#define SHAPEL(T, F64, X, Y, Z) \
    if (f64 == F64 && x == X && y == Y && z == Z) return go<T, X, Y, Z, true>(a, s, query);
    // Measured against what ran before, 1 GiB per side | the reference's 32 MiB (profiles/r06_k_planes_halves.log): (16, 16, 128) 0.446 (one
    // tile per CU) -> 0.606 | 0.270 -> 0.422; the two-launch shapes 0.31-0.39 -> 0.46-0.62 | 0.28-0.38 -> 0.31-0.43 -- 0.89-1.09 of their
    // interleaved twins at 1 GiB; numpy (64, 512) loses at 32 MiB (0.361 -> 0.332): LARGE_ONLY
    // (16, 16, 128): a one-launch plan in the split layout too (fft_nd2p.hip holds its one-tile kernel for in-place executes)
    SHAPE(float, 0, 128, 16, 16)
    // the other 32768-point shapes of fft_nd2z_f32.hip: TWO launches as split-complex plans; out of place one launch (the plan's _oop_nd)
    SHAPE(float, 0, 64, 64, 8) SHAPE(float, 0, 128, 256, 1) SHAPE(float, 0, 256, 128, 1) SHAPE(float, 0, 64, 512, 1) SHAPEL(float, 0, 512, 64, 1)
    SHAPE(float, 0, 2048, 16, 1) SHAPE(float, 0, 16, 2048, 1) SHAPE(float, 0, 64, 8, 64) SHAPE(float, 0, 16, 128, 16) SHAPE(float, 0, 16, 16, 128)
    SHAPE(float, 0, 64, 32, 16) SHAPE(float, 0, 64, 16, 32) SHAPE(float, 0, 32, 64, 16) SHAPE(float, 0, 32, 16, 64) SHAPE(float, 0, 16, 64, 32)
#define SHAPE4(T, F64, X, Y, Z) \
    if (f64 == F64 && x == X && y == Y && z == Z) return go4<T, X, Y, Z>(a, s, query);
    SHAPE4(float, 0, 256, 256, 1) SHAPE4(float, 0, 128, 512, 1) SHAPE4(float, 0, 64, 1024, 1) SHAPE4(float, 0, 1024, 64, 1) SHAPE4(float, 0, 64, 64, 16)
#undef SHAPE4
#undef SHAPE
#undef SHAPEL
    return -2;
}

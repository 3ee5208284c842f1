// fp64 instances of the pass-pair kernels (fft_pair.hpp): 256 x 256 planes with 256 z (BASELINE config 4, both layouts).
// Interleaved: the y axis split 32 x 8 (XY tile 128 KiB at two work-groups per CU, YZ tile 256 KiB at one) or 64 x 4 (the other
// way round; measured 28.7 against 30.7 %, profiles/r03_b_c4_pair_split.log; the 256 KiB YZ tile on 512 threads x 32 points
// instead of 1024 x 16, and non-temporal loads of the intermediate, both measured within 1 % of this form on bench.py --config c4).  Split planes: 64 x 4 with 16-column YZ tiles
// (16 x 8 bytes = one 128-byte segment per plane); the intermediate between the two launches is interleaved either way.
#include "mifft_internal.h"
#include "fft_pair.hpp"

using namespace mifft;

// kind 0 = XY keyed by (nx, R0, R1); kind 1 = YZ keyed by (S0, R1, nz); `split`: XY reads / YZ writes two scalar planes.
// 0 = launched (query: exists), -2 = no such kernel.  *width = adjacent columns per YZ tile (tiles = batch * S0 / width).
extern "C" int mifft_pair_f64(int kind, int k0, int k1, int k2, int split, const PairArgs* a, hipStream_t s, int query, int* width) {
#define XY(NX, R0, R1, NT, HALF, OCC, RLX, RLY, SPLIT)                                       \
    if (kind == 0 && k0 == NX && k1 == R0 && k2 == R1 && split == (SPLIT ? 1 : 0))          \
        return query ? 0 : launch_pair<double, PairXY<double, NX, R0, R1, NT, HALF, OCC, RLX, RLY, SPLIT>>(a, s);
#define YZ(S0, R1, NZ, W, NT, HALF, OCC, RLY, RLZ, SPLIT)                                    \
    if (kind == 1 && k0 == S0 && k1 == R1 && k2 == NZ && split == (SPLIT ? 1 : 0)) {        \
        if (width) *width = W;                                                               \
        return query ? 0 : launch_pair<double, PairYZ<double, S0, R1, NZ, W, NT, HALF, OCC, RLY, RLZ, SPLIT>>(a, s); \
    }
#define RL(...) RadixList<__VA_ARGS__>
    XY(256, 32, 8, 512, true, 4, RL(16, 16), RL(8, 4), false)
    YZ(256 * 32, 8, 256, 8, 1024, true, 4, RL(8), RL(16, 16), false)
    XY(256, 64, 4, 1024, true, 4, RL(16, 16), RL(16, 4), false)
    YZ(256 * 64, 4, 256, 8, 512, true, 4, RL(4), RL(16, 16), false)
    // 128^3: y = 32 x 4; XY tile 4096 points (64 KiB), YZ tile 8 x 4 x 128 = 4096 points (64 KiB)
    XY(128, 32, 4, 256, false, 1, RL(8, 16), RL(8, 4), false)
    YZ(128 * 32, 4, 128, 8, 256, false, 1, RL(4), RL(8, 16), false)
    XY(256, 64, 4, 1024, true, 4, RL(16, 16), RL(16, 4), true)
    YZ(256 * 64, 4, 256, 16, 1024, true, 4, RL(4), RL(16, 16), true)
    // Round 5 (see fft_pair_f32.hip): 2-D shapes with a 4096-point y axis -- XY on NX x R0 = 8192 points --, 3-D shapes with short y and z
    // behind a long x -- YZ on 8 adjacent x (128-byte segments) x whole (z, y) planes
    XY(256, 32, 128, 512, true, 4, RL(16, 16), RL(8, 4), false)
    XY(512, 16, 256, 512, true, 4, RL(2, 16, 16), RL(16), false)
    XY(1024, 8, 512, 512, true, 4, RL(4, 16, 16), RL(8), false)
    XY(2048, 4, 1024, 512, true, 4, RL(8, 16, 16), RL(4), false)
    XY(128, 64, 64, 512, true, 4, RL(8, 16), RL(16, 4), false)          // (4096, 128): 0.231 on three launches (profiles/r05_shape_grid_survey.log)
    XY(4096, 2, 2048, 512, true, 4, RL(16, 16, 16), RL(2), false)       // (4096, 4096): 0.211
    YZ(1024, 32, 32, 8, 512, true, 4, RL(8, 4), RL(8, 4), false)
    YZ(2048, 32, 32, 8, 512, true, 4, RL(8, 4), RL(8, 4), false)
    YZ(1024, 16, 16, 8, 256, false, 1, RL(8, 2), RL(8, 2), false)
    YZ(2048, 16, 16, 8, 256, false, 1, RL(8, 2), RL(8, 2), false)
    // Round 5, after the shape survey (profiles/r05_shape_grid_survey.log: fp64 3-D shapes with 256-point x rows next to a shorter axis ran
    // THREE launches, 0.20-0.26): pass pairs for (z, y, 256) with y in {128, 256} and z in {64, 128, 256} -- y = 32 x 8 or 32 x 4, the XY tile
    // of 256^3, YZ tiles of 2048 ... 8192 points on 8 adjacent x
    XY(256, 32, 4, 512, true, 4, RL(16, 16), RL(8, 4), false)
    YZ(256 * 32, 8, 128, 8, 512, true, 4, RL(8), RL(8, 16), false)
    YZ(256 * 32, 8, 64, 8, 256, false, 1, RL(8), RL(8, 8), false)
    YZ(256 * 32, 4, 256, 8, 512, true, 4, RL(4), RL(16, 16), false)
    YZ(256 * 32, 4, 128, 8, 256, false, 1, RL(4), RL(8, 16), false)
    YZ(256 * 32, 4, 64, 8, 256, false, 1, RL(4), RL(8, 8), false)
    // ... z = 32 behind 256-point rows, and (z, 256, 128) with z in {32 ... 256}: 128^3's XY tile, YZ tiles of 2048 ... 16384 points
    YZ(256 * 32, 8, 32, 8, 256, false, 1, RL(8), RL(8, 4), false)
    YZ(256 * 32, 4, 32, 8, 128, false, 1, RL(4), RL(8, 4), false)
    XY(128, 32, 8, 256, false, 1, RL(8, 16), RL(8, 4), false)
    YZ(128 * 32, 8, 256, 8, 1024, true, 4, RL(8), RL(16, 16), false)
    YZ(128 * 32, 8, 128, 8, 512, true, 4, RL(8), RL(8, 16), false)
    YZ(128 * 32, 8, 64, 8, 256, false, 1, RL(8), RL(8, 8), false)
    YZ(128 * 32, 8, 32, 8, 256, false, 1, RL(8), RL(8, 4), false)
    // ... and (z, 256, 64): XY tile of 2048 points
    XY(64, 32, 8, 256, false, 1, RL(8, 8), RL(8, 4), false)
    YZ(64 * 32, 8, 256, 8, 1024, true, 4, RL(8), RL(16, 16), false)
    YZ(64 * 32, 8, 128, 8, 512, true, 4, RL(8), RL(8, 16), false)
    YZ(64 * 32, 8, 64, 8, 256, false, 1, RL(8), RL(8, 8), false)
    YZ(64 * 32, 8, 32, 8, 256, false, 1, RL(8), RL(8, 4), false)
#undef XY
#undef YZ
#undef RL
    return -2;
}

// fp32 instances of the pass-pair kernels (fft_pair.hpp): 256^3 interleaved with the y axis split 64 x 4 -- both tiles are
// 16384 points = 128 KiB (two work-groups per CU in half-exchange form), YZ tiles 16 columns wide (128-byte segments).
#include "mifft_internal.h"
#include "fft_pair.hpp"

using namespace mifft;

extern "C" int mifft_pair_f32(int kind, int k0, int k1, int k2, int split, const PairArgs* a, hipStream_t s, int query, int* width) {
#define XY(NX, R0, R1, NT, HALF, OCC, RLX, RLY, SPLIT)                                       \
    if (kind == 0 && k0 == NX && k1 == R0 && k2 == R1 && split == (SPLIT ? 1 : 0))          \
        return query ? 0 : launch_pair<float, PairXY<float, NX, R0, R1, NT, HALF, OCC, RLX, RLY, SPLIT>>(a, s);
#define YZ(S0, R1, NZ, W, NT, HALF, OCC, RLY, RLZ, SPLIT)                                    \
    if (kind == 1 && k0 == S0 && k1 == R1 && k2 == NZ && split == (SPLIT ? 1 : 0)) {        \
        if (width) *width = W;                                                               \
        return query ? 0 : launch_pair<float, PairYZ<float, S0, R1, NZ, W, NT, HALF, OCC, RLY, RLZ, SPLIT>>(a, s); \
    }
#define RL(...) RadixList<__VA_ARGS__>
    XY(256, 64, 4, 512, true, 4, RL(16, 16), RL(16, 4), false)
    YZ(256 * 64, 4, 256, 16, 512, true, 4, RL(4), RL(16, 16), false)
    // 128^3: y = 32 x 4; XY tile 4096 points (32 KiB; half-form exchange), YZ tile 16 x 4 x 128 = 8192 points (64 KiB) -- against one 16384-point plane
    // tile + a generic 128-point column pass
    // (round 4: the y digit as one radix-32 stage on 128 threads x 32 points -- four LDS exchanges per point instead of five; the
    // persistent two-pair kernel runs the same list, fft_fusedp.hip)
    XY(128, 32, 4, 128, true, 1, RL(8, 16), RL(32), false)
    YZ(128 * 32, 4, 128, 16, 512, false, 1, RL(4), RL(8, 16), false)
    // Round 5: pairs as the way out of a THIRD launch.  (a) 2-D shapes with a 4096-point y axis (two strided passes 64 x 64 after the row
    // pass: three launches, (4096, 256) 0.257 of the roofline while its transpose runs 0.384): XY = ROW x + COL y (R0) on NX x R0 = 16384
    // points, then ONE plain strided pass of R1 points.  (b) 3-D shapes with short y and z behind a long x ((32, 32, 2048): row + col y +
    // col z 0.264, its transpose 0.355): YZ = COL y (all of it) + COL z on 16 adjacent x, whole (z, y) planes of <= 1024 points.
    XY(128, 128, 32, 512, true, 4, RL(8, 16), RL(16, 8), false)         // (4096, 128): 0.265 on three launches (profiles/r05_shape_grid_survey.log)
    XY(256, 64, 64, 512, true, 4, RL(16, 16), RL(16, 4), false)
    XY(512, 32, 128, 512, true, 4, RL(2, 16, 16), RL(8, 4), false)
    XY(1024, 16, 256, 512, true, 4, RL(4, 16, 16), RL(16), false)
    XY(2048, 8, 512, 512, true, 4, RL(8, 16, 16), RL(8), false)
    XY(4096, 4, 1024, 512, true, 4, RL(16, 16, 16), RL(4), false)
    YZ(2048, 32, 32, 16, 512, true, 4, RL(8, 4), RL(8, 4), false)
    YZ(4096, 32, 32, 16, 512, true, 4, RL(8, 4), RL(8, 4), false)
    YZ(2048, 16, 16, 16, 256, false, 1, RL(16), RL(16), false)
    YZ(4096, 16, 16, 16, 256, false, 1, RL(16), RL(16), false)
    // (c) after the shape survey (profiles/r05_shape_grid_survey.log): (z, 256, 256) with z in {64, 128} on the two pairs of 256^3 -- y = 64 x 4,
    // its XY tile, YZ tiles of 4096 / 8192 points on 16 adjacent x -- instead of three launches: 0.245 -> 0.329, 0.269 -> 0.321 at 1 GiB.
    // (The y = 128 shapes -- 64 x 2 -- already run two launches, the 32768-point (y, x) plane as one tile + the z pass: the pairs measured
    // + 1 ... + 5 % at 1 GiB and - 8 % at 32 MiB against them; not instantiated.  profiles/r05_pass_pairs_256_point_rows.log)
    YZ(256 * 64, 4, 128, 16, 512, false, 1, RL(4), RL(8, 16), false)
    YZ(256 * 64, 4, 64, 16, 256, false, 1, RL(4), RL(8, 8), false)
    YZ(256 * 64, 4, 32, 16, 256, false, 1, RL(4), RL(8, 4), false)      // (32, 256, 256): 0.271 on three launches
#undef XY
#undef YZ
#undef RL
    return -2;
}

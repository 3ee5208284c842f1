// fp64 instances of the pass-pair kernels (fft_pair.hpp): 256 x 256 planes with 256 z (BASELINE config 4), the y axis split
// 32 x 8 (XY tile 128 KiB at two work-groups per CU, YZ tile 256 KiB at one) or 64 x 4 (the other way round).
#include "mifft_internal.h"
#include "fft_pair.hpp"

using namespace mifft;

// kind 0 = XY keyed by (nx, R0, R1); kind 1 = YZ keyed by (S0, R1, nz).  0 = launched (query: exists), -2 = no such kernel.
extern "C" int mifft_pair_f64(int kind, int k0, int k1, int k2, const PairArgs* a, hipStream_t s, int query) {
#define XY(NX, R0, R1, NT, HALF, OCC, RLX, RLY)                                              \
    if (kind == 0 && k0 == NX && k1 == R0 && k2 == R1)                                       \
        return query ? 0 : launch_pair<double, PairXY<double, NX, R0, R1, NT, HALF, OCC, RLX, RLY>>(a, s);
#define YZ(S0, R1, NZ, NT, HALF, OCC, RLY, RLZ)                                              \
    if (kind == 1 && k0 == S0 && k1 == R1 && k2 == NZ)                                       \
        return query ? 0 : launch_pair<double, PairYZ<double, S0, R1, NZ, 8, NT, HALF, OCC, RLY, RLZ>>(a, s);
#define RL(...) RadixList<__VA_ARGS__>
    XY(256, 32, 8, 512, true, 4, RL(16, 16), RL(8, 4))
    YZ(256 * 32, 8, 256, 1024, true, 4, RL(8), RL(16, 16))
    XY(256, 64, 4, 1024, true, 4, RL(16, 16), RL(16, 4))
    YZ(256 * 64, 4, 256, 512, true, 4, RL(4), RL(16, 16))
#undef XY
#undef YZ
#undef RL
    return -2;
}

// fp64 instances of the fixed-shape N-D kernel (fft_nd2.hpp): every 2-D shape with both axes in 16...512 that fits one
// tile (x*y <= 8192) and the common small 3-D shapes; configuration per shape by Nd2Auto.  Every other shape runs on the
// run-time-shaped kernel of fft_nd.hpp.
#include "mifft_internal.h"
#include "fft_nd2.hpp"

namespace {
using namespace mifft;
int launch_shape(int x, int y, int z, const TileArgs* a, hipStream_t s, int query) {
#define SHAPE(X, Y, Z)                                          \
    if (x == X && y == Y && z == Z) {                           \
        if (query) return 0;                                    \
        return launch_nd2_auto<double, X, Y, Z>(a, s);           \
    }
    SHAPE(16, 16, 1) SHAPE(16, 32, 1) SHAPE(16, 64, 1) SHAPE(16, 128, 1)
    SHAPE(16, 256, 1) SHAPE(16, 512, 1) SHAPE(32, 16, 1) SHAPE(32, 32, 1)
    SHAPE(32, 64, 1) SHAPE(32, 128, 1) SHAPE(32, 256, 1) SHAPE(64, 16, 1)
    SHAPE(64, 32, 1) SHAPE(64, 64, 1) SHAPE(64, 128, 1) SHAPE(128, 16, 1)
    SHAPE(128, 32, 1) SHAPE(128, 64, 1) SHAPE(256, 16, 1) SHAPE(256, 32, 1)
    SHAPE(512, 16, 1) SHAPE(8, 8, 1) SHAPE(16, 16, 16) SHAPE(64, 8, 8)
    SHAPE(8, 8, 8) SHAPE(32, 16, 16) SHAPE(16, 16, 8) SHAPE(16, 16, 32)
    SHAPE(32, 32, 8) SHAPE(16, 8, 8)
    // one-tile shapes beyond the run-time-shaped kernel's largest tile (interleaved data only)
    SHAPE(32, 512, 1) SHAPE(64, 256, 1) SHAPE(128, 128, 1) SHAPE(256, 64, 1)
    SHAPE(512, 32, 1) SHAPE(32, 32, 16) SHAPE(64, 16, 16) SHAPE(16, 16, 64)
#undef SHAPE
    return -2;
}
}  // namespace

// 0 when a kernel for the (x, y, z) shape exists
extern "C" int mifft_nd2_f64_supported(int x, int y, int z) { return launch_shape(x, y, z, nullptr, nullptr, 1); }

extern "C" int mifft_nd2_f64_launch(int x, int y, int z, const mifft::TileArgs* a, hipStream_t s) {
    return launch_shape(x, y, z, a, s, 0);
}

// fp32 instances of the fixed-shape N-D kernel (fft_nd2.hpp) for the common small 2-D / 3-D shapes (the reference's
// published table, doc/source/index.rst:357-373, has (16,16), (128,128), (16,16,16), and the planes of (32,32,128) and
// (128,128,128)); every other shape runs on the run-time-shaped kernel of fft_nd.hpp.
#include "mifft_internal.h"
#include "fft_nd2.hpp"

namespace {
using namespace mifft;
// (x, y, z) -> launcher
int launch_shape(int x, int y, int z, const TileArgs* a, hipStream_t s, int query) {
#define SHAPE(X, Y, Z, ...)                      \
    if (x == X && y == Y && z == Z) {            \
        if (query) return 0;                     \
        return launch_nd2<float, X, Y, Z, __VA_ARGS__>(a, s); \
    }
    //    shape          P     NT  HALF OCC EDGE_IN  radices x, y, z
    SHAPE(16, 16, 1,     4096, 256, false, 1, false, RadixList<16>, RadixList<16>)
    SHAPE(32, 32, 1,     4096, 256, false, 1, true,  RadixList<2, 16>, RadixList<16, 2>)
    SHAPE(64, 64, 1,     4096, 256, false, 1, true,  RadixList<4, 16>, RadixList<16, 4>)
    SHAPE(128, 16, 1,    4096, 256, false, 1, true,  RadixList<8, 16>, RadixList<16>)
    SHAPE(128, 32, 1,    4096, 256, false, 1, true,  RadixList<8, 16>, RadixList<16, 2>)
    SHAPE(128, 64, 1,    8192, 512, false, 1, true,  RadixList<8, 16>, RadixList<16, 4>)
    SHAPE(64, 128, 1,    8192, 512, false, 1, true,  RadixList<4, 16>, RadixList<16, 8>)
    SHAPE(128, 128, 1,   16384, 512, true, 4, true,  RadixList<8, 16>, RadixList<16, 8>)
    SHAPE(16, 16, 16,    4096, 256, false, 1, false, RadixList<16>, RadixList<16>, RadixList<16>)
    SHAPE(64, 8, 8,      4096, 256, false, 1, true,  RadixList<4, 16>, RadixList<8>, RadixList<8>)
#undef SHAPE
    return -2;
}
}  // namespace

// 0 when a kernel for the (x, y, z) shape exists
extern "C" int mifft_nd2_f32_supported(int x, int y, int z) { return launch_shape(x, y, z, nullptr, nullptr, 1); }

extern "C" int mifft_nd2_f32_launch(int x, int y, int z, const mifft::TileArgs* a, hipStream_t s) {
    return launch_shape(x, y, z, a, s, 0);
}

// Dense split-complex N-D shapes on 16-byte plane accesses (fft_nd2p.hpp, round 6): the reference's published N-D shapes
// (doc/source/index.rst:357-373: (16, 16), (128, 128), (8, 8, 64), (16, 16, 16), (16, 16, 128)) and their neighbours, float32 and float64.
#include "../../include/mifft.h"
#include "mifft_internal.h"
#include "fft_nd2p.hpp"

using namespace mifft;

// query != 0: 0 if the shape has an instance, -2 if not (nothing is launched)
extern "C" int mifft_nd2p(int f64, int x, int y, int z, const TileArgs* a, hipStream_t s, int query) {
#define SHAPE(T, X, Y, Z) \
    if (x == X && y == Y && z == Z) return query ? 0 : launch_nd2p_auto<T, X, Y, Z>(a, s);
    if (!f64) {
        SHAPE(float, 16, 16, 1) SHAPE(float, 32, 32, 1) SHAPE(float, 64, 64, 1) SHAPE(float, 128, 128, 1)
        SHAPE(float, 16, 16, 16) SHAPE(float, 64, 8, 8) SHAPE(float, 128, 16, 16) SHAPE(float, 32, 32, 32)
    } else {
        SHAPE(double, 16, 16, 1) SHAPE(double, 32, 32, 1) SHAPE(double, 64, 64, 1) SHAPE(double, 128, 128, 1)
        SHAPE(double, 16, 16, 16) SHAPE(double, 64, 8, 8)
    }
#undef SHAPE
    return -2;
}

// Instances of the tiled fixed-shape N-D kernel (fft_nd2t.hpp): the tile shapes a "tiled batch" plan (pyfft_amd/generic.py,
// Plan(tile, parent_shape=...)) runs in ONE launch straight on the parent array; other tile shapes keep the gather / dense plan /
// scatter form.  Squares 8 ... 128 and cubes 8 ... 32 (fp64: cubes to 16), plus a few rectangles.
#include "mifft_internal.h"
#include "fft_nd2t.hpp"

using namespace mifft;

#ifndef MIFFT_ND2T_SPLIT
#define MIFFT_ND2T_SPLIT false
#define MIFFT_ND2T_NAME mifft_nd2t
#endif
#ifndef MIFFT_ND2T_SPLIT_OUT
#define MIFFT_ND2T_SPLIT_OUT MIFFT_ND2T_SPLIT
#endif

extern "C" int MIFFT_ND2T_NAME(int f64, int x, int y, int z, const TileArgs* a, const TiledGeom* g, hipStream_t s, int query) {
#define SHAPE(T, F, X, Y, Z)                                         \
    if (f64 == F && x == X && y == Y && z == Z)                      \
        return query ? 0 : launch_nd2t_auto<T, X, Y, Z, MIFFT_ND2T_SPLIT, MIFFT_ND2T_SPLIT_OUT>(a, g, s);
#define BOTH(X, Y, Z) SHAPE(float, 0, X, Y, Z) SHAPE(double, 1, X, Y, Z)
    BOTH(8, 8, 1) BOTH(16, 16, 1) BOTH(32, 32, 1) BOTH(64, 64, 1) BOTH(128, 128, 1)
    BOTH(32, 16, 1) BOTH(64, 32, 1) BOTH(128, 64, 1)
    BOTH(8, 8, 8) BOTH(16, 16, 16)
    SHAPE(float, 0, 32, 32, 32) SHAPE(double, 1, 32, 32, 16)
    BOTH(16, 16, 8) BOTH(32, 32, 8)
#undef BOTH
#undef SHAPE
    return -2;
}

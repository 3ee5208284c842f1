// Dense split-complex N-D shapes on 16-byte plane accesses (fft_nd2p.hpp, round 6): the reference's published N-D shapes
// (doc/source/index.rst:357-373: (16, 16), (128, 128), (8, 8, 64), (16, 16, 16), (16, 16, 128)) and their neighbours, float32 and float64.
// Which shape runs it at which launch size is what profiles/r06_f_planes_probe.log measured against the kernels it replaces (the tiled
// fixed-shape kernel with one tile per parent, the run-time-shaped kernel, or two launches): ALL = every size, BIG = only launches beyond
// the write-through size (half the last-level cache per side) -- in small launches the extra exchange on either side of the stages is
// latency the tile cannot hide.  (128, 128) lost or tied at both sizes in both precisions (fp32 0.637 / 0.409 against 0.684 / 0.440) and
// has no instance.
#include "../../include/mifft.h"
#include "mifft_internal.h"
#include "fft_nd2p.hpp"

using namespace mifft;

// query 1: 0 if the shape has an instance; query 2: 0 if that instance is also the choice for SMALL launches; else -2.  query 0 launches.
extern "C" int mifft_nd2p(int f64, int x, int y, int z, const TileArgs* a, hipStream_t s, int query) {
#define ALL(T, X, Y, Z) \
    if (x == X && y == Y && z == Z) return query ? 0 : launch_nd2p_auto<T, X, Y, Z>(a, s);
#define BIG(T, X, Y, Z) \
    if (x == X && y == Y && z == Z) return query ? (query == 2 ? -2 : 0) : launch_nd2p_auto<T, X, Y, Z>(a, s);
    if (!f64) {
        // 1 GiB / 32 MiB per side, fraction of the roofline, this kernel against what ran before:
        ALL(float, 16, 16, 1)      // 0.756 / 1.024 against 0.735 / 0.835
        ALL(float, 32, 32, 1)      // 0.689 / 0.633 against 0.487 / 0.399 (the run-time-shaped kernel)
        ALL(float, 64, 64, 1)      // 0.735 / 0.692 against 0.625 / 0.575
        ALL(float, 16, 16, 16)     // 0.775 / 0.865 against 0.700 / 0.645
        ALL(float, 64, 8, 8)       // 0.775 / 0.770 against 0.587 / 0.546
        ALL(float, 128, 16, 16)    // 0.446 / 0.270 against 0.353 / 0.302 (two launches): one chain per plan, the big launches decide
        ALL(float, 32, 32, 32)     // 0.519 / 0.326 against 0.298 / 0.265 (two launches)
    } else {
        ALL(double, 16, 16, 1)     // 0.769 / 0.726 against 0.689 / 0.643
        BIG(double, 32, 32, 1)     // 0.776 / 0.714 against 0.724 / 0.783
        BIG(double, 64, 64, 1)     // 0.734 / 0.671 against 0.716 / 0.747
        BIG(double, 16, 16, 16)    // 0.638 / 0.552 against 0.603 / 0.559
        ALL(double, 64, 8, 8)      // 0.770 / 0.654 against 0.604 / 0.392
    }
#undef ALL
#undef BIG
    return -2;
}

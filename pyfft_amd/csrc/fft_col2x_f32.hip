// fp32 split-complex instances of the lane-interleaved double tile as plain launches (fft_col2.hpp fft_col2x_kernel): the
// transposing first pass of a long axis reading the planes (its output is the plan's interleaved temp) and the plain strided pass
// writing them.  -fno-slp-vectorize: see fft_col2_f32.hip.
#include "mifft_internal.h"
#include "fft_col2.hpp"

namespace {
template <int A> int launch(int tr, const mifft::TileArgs* a, hipStream_t s) {
    const long long tiles = a->total / 32;
    if (tiles > 2147483647ll) return -1;
    const dim3 g((unsigned)tiles), b(512);
    if (tr) hipLaunchKernelGGL((mifft::fft_col2x_kernel<A, true, true, true, false>), g, b, 0, s, *a);
    else hipLaunchKernelGGL((mifft::fft_col2x_kernel<A, false, false, false, true>), g, b, 0, s, *a);
    return (int)hipGetLastError();
}
}  // namespace

// 1 = this pass has such a kernel: planes on exactly the HBM side (in for the transposing pass, out for the plain one), 32 whole
// columns of one matrix per tile
extern "C" int mifft_col2x_f32_eligible(int L, int tr, const mifft::TileArgs* a) {
    if (L != 256 && L != 512 && L != 1024) return 0;
    if (mifft_debug_get(MIFFT_DEBUG_NARROW_TILES) == 1) return 0;
    if (tr ? !(a->split && !a->split_out) : !(!a->split && a->split_out)) return 0;
    // the pass that WRITES planes gains from whole lines only while two such work-groups share a CU (L <= 512): 1024 x 1024 x 16 planes
    // 0.336 on the 16-column tiles, 0.319 here (partial-line WRITES cost no second crossing); the pass that READS them always gains
    // (profiles/r04_aq_split_chain_mode.log)
    if (!tr && L == 1024 && mifft_debug_get(MIFFT_DEBUG_NARROW_TILES) != 2) return 0;
    return (a->total & 31) == 0 && a->logMS >= 5;
}

extern "C" int mifft_col2x_f32_launch(int L, int tr, const mifft::TileArgs* a, hipStream_t s) {
    switch (L) {
        case 1024: return launch<4>(tr, a, s);
        case 512: return launch<2>(tr, a, s);
        case 256: return launch<1>(tr, a, s);
    }
    return -2;
}

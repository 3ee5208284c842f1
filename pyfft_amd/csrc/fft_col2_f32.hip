// fp32 two-phase strided-axis kernels (fft_col2.hpp) for L = 256 / 512 / 1024.
// Built with -fno-slp-vectorize: the SLP vectoriser turns the complex multiplies into v_pk_fma_f32 pairs that
// use only half of each 64-bit result, which doubles the live registers of the 64-point-per-thread kernel and
// makes it spill (measured: 228 bytes/lane of scratch with SLP, 36 without).
#include "mifft_internal.h"
#include "fft_col2.hpp"

namespace {
template <int A, bool TR, bool TW> int launch_l(const mifft::TileArgs* a, hipStream_t s) {
    const long long tiles = a->total / 16;
    if (tiles > 2147483647ll) return -1;
    const dim3 g((unsigned)tiles), b(256);
    if (a->split && a->split_out)
        hipLaunchKernelGGL((mifft::fft_col2_kernel<float, A, TR, TW, true, true>), g, b, 0, s, *a);
    else if (a->split)
        hipLaunchKernelGGL((mifft::fft_col2_kernel<float, A, TR, TW, true, false>), g, b, 0, s, *a);
    else if (a->split_out)
        hipLaunchKernelGGL((mifft::fft_col2_kernel<float, A, TR, TW, false, true>), g, b, 0, s, *a);
    else
        hipLaunchKernelGGL((mifft::fft_col2_kernel<float, A, TR, TW, false, false>), g, b, 0, s, *a);
    return (int)hipGetLastError();
}
// Instances: the transposing first pass of a long axis (S == 1, always with the inter-pass twiddle) and the plain
// strided pass (M == 1, no twiddle).  Other combinations use the generic tile kernel.
template <int A> int launch(int tr, const mifft::TileArgs* a, hipStream_t s) {
    return tr ? launch_l<A, true, true>(a, s) : launch_l<A, false, false>(a, s);
}
}  // namespace

// The tile must be 16 whole columns of one matrix (M*S >= 16) and the per-thread 32-bit byte offsets must fit.
extern "C" int mifft_col2_f32_eligible(int L, int tr, const mifft::TileArgs* a) {
    if (L != 256 && L != 512 && L != 1024) return 0;
    if (a->total <= 0 || a->logMS < 4 || a->logMS > 24 || a->logS > 22) return 0;
    return tr ? (a->has_tw != 0) : (a->has_tw == 0);
}

extern "C" int mifft_col2_f32_launch(int L, int tr, const mifft::TileArgs* a, hipStream_t s) {
    switch (L) {
        case 1024: return launch<4>(tr, a, s);
        case 512: return launch<2>(tr, a, s);
        case 256: return launch<1>(tr, a, s);
    }
    return -2;
}

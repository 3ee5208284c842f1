// fp32 instances of the 512-thread L = 2048 strided-axis kernel (fft_col3.hpp).  -fno-slp-vectorize: see fft_col2_f32.hip.
#include "mifft_internal.h"
#include "fft_col3.hpp"

namespace {
template <bool TR, bool TW> int launch(const mifft::TileArgs* a, hipStream_t s) {
    const long long tiles = a->total / 16;
    if (tiles > 2147483647ll) return -1;
    const dim3 g((unsigned)tiles), b(512);
    if (a->split && a->split_out)
        hipLaunchKernelGGL((mifft::fft_col3_kernel<float, 4, TR, TW, true, true>), g, b, 0, s, *a);
    else if (a->split)
        hipLaunchKernelGGL((mifft::fft_col3_kernel<float, 4, TR, TW, true, false>), g, b, 0, s, *a);
    else if (a->split_out)
        hipLaunchKernelGGL((mifft::fft_col3_kernel<float, 4, TR, TW, false, true>), g, b, 0, s, *a);
    else
        hipLaunchKernelGGL((mifft::fft_col3_kernel<float, 4, TR, TW, false, false>), g, b, 0, s, *a);
    return (int)hipGetLastError();
}
}  // namespace

// The tile must be 16 whole columns of one matrix (M*S >= 16) and the per-thread 32-bit byte offsets must fit:
// input (31 << logMS) * 8 and output (1039 << logS) * 8 bytes.
extern "C" int mifft_col3_f32_eligible(int L, int tr, const mifft::TileArgs* a) {
    if (L != 2048) return 0;
    if (a->total <= 0 || (a->total & 15) || a->logMS < 4 || a->logMS > 23 || a->logS > 18) return 0;
    return tr ? (a->has_tw != 0) : (a->has_tw == 0);
}

extern "C" int mifft_col3_f32_launch(int tr, const mifft::TileArgs* a, hipStream_t s) {
    return tr ? launch<true, true>(a, s) : launch<false, false>(a, s);
}

// fp64 instances of the 512-thread strided-axis kernel (fft_col3.hpp): L = 1024.  -fno-slp-vectorize: see fft_col2_f32.hip.
#include "mifft_internal.h"
#include "fft_col3.hpp"

namespace {
template <bool TR, bool TW> int launch(const mifft::TileArgs* a, hipStream_t s) {
    const long long tiles = a->total / 16;
    if (tiles > 2147483647ll) return -1;
    const dim3 g((unsigned)tiles), b(512);
    if (a->split && a->split_out)
        hipLaunchKernelGGL((mifft::fft_col3_kernel<double, 2, TR, TW, true, true>), g, b, 0, s, *a);
    else if (a->split)
        hipLaunchKernelGGL((mifft::fft_col3_kernel<double, 2, TR, TW, true, false>), g, b, 0, s, *a);
    else if (a->split_out)
        hipLaunchKernelGGL((mifft::fft_col3_kernel<double, 2, TR, TW, false, true>), g, b, 0, s, *a);
    else
        hipLaunchKernelGGL((mifft::fft_col3_kernel<double, 2, TR, TW, false, false>), g, b, 0, s, *a);
    return (int)hipGetLastError();
}
}  // namespace

// 16 whole columns of one matrix and 32-bit per-thread byte offsets: input (31 << logMS) * 16, output (527 << logS) * 16 bytes
extern "C" int mifft_col3_f64_eligible(int L, int tr, const mifft::TileArgs* a) {
    if (L != 1024) return 0;
    if (a->total <= 0 || (a->total & 15) || a->logMS < 4 || a->logMS > 22 || a->logS > 18) return 0;
    return tr ? (a->has_tw != 0) : (a->has_tw == 0);
}

extern "C" int mifft_col3_f64_launch(int tr, const mifft::TileArgs* a, hipStream_t s) {
    return tr ? launch<true, true>(a, s) : launch<false, false>(a, s);
}

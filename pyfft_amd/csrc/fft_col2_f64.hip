// fp64 two-phase strided-axis kernels (fft_col2.hpp) for L = 256 and 512 (16 / 32 points per thread, one 68 KiB LDS
// exchange buffer, two work-groups per CU; 16 columns = 256-byte segments interleaved, 128-byte segments per plane
// when split).  L = 1024 in fp64 would need 256 data VGPRs per thread and stays on the generic tile kernel.
#include "mifft_internal.h"
#include "fft_col2.hpp"

namespace {
template <int A, bool TR, bool TW> int launch_l(const mifft::TileArgs* a, hipStream_t s) {
    const long long tiles = a->total / 16;
    if (tiles > 2147483647ll) return -1;
    const dim3 g((unsigned)tiles), b(256);
    if (a->split && a->split_out)
        hipLaunchKernelGGL((mifft::fft_col2_kernel<double, A, TR, TW, true, true>), g, b, 0, s, *a);
    else if (a->split)
        hipLaunchKernelGGL((mifft::fft_col2_kernel<double, A, TR, TW, true, false>), g, b, 0, s, *a);
    else if (a->split_out)
        hipLaunchKernelGGL((mifft::fft_col2_kernel<double, A, TR, TW, false, true>), g, b, 0, s, *a);
    else
        hipLaunchKernelGGL((mifft::fft_col2_kernel<double, A, TR, TW, false, false>), g, b, 0, s, *a);
    return (int)hipGetLastError();
}
}  // namespace

extern "C" int mifft_col2_f64_eligible(int L, int tr, const mifft::TileArgs* a) {
    if (L != 256 && L != 512) return 0;
    if (a->total <= 0 || a->logMS < 4 || a->logMS > 23 || a->logS > 21) return 0;
    return tr ? (a->has_tw != 0) : (a->has_tw == 0);
}

extern "C" int mifft_col2_f64_launch(int L, int tr, const mifft::TileArgs* a, hipStream_t s) {
    if (L == 256) return tr ? launch_l<1, true, true>(a, s) : launch_l<1, false, false>(a, s);
    if (L == 512) return tr ? launch_l<2, true, true>(a, s) : launch_l<2, false, false>(a, s);
    return -2;
}

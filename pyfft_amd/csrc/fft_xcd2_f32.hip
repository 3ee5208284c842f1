// fp32 instances of the XCD-cooperative single-crossing kernel (fft_xcd2.hpp).  -fno-slp-vectorize: see fft_col2_f32.hip.
#include "mifft_internal.h"
#include "fft_xcd2.hpp"

extern "C" int mifft_xcd2_f32_launch(const mifft::Xcd2Args* f, int split, int prefetch, unsigned grid, hipStream_t s) {
    if (split) {
        if (prefetch) hipLaunchKernelGGL((mifft::fft_xcd2_kernel<float, true, true, true>), dim3(grid), dim3(256), 0, s, *f);
        else hipLaunchKernelGGL((mifft::fft_xcd2_kernel<float, true, true, false>), dim3(grid), dim3(256), 0, s, *f);
    } else {
        if (prefetch) hipLaunchKernelGGL((mifft::fft_xcd2_kernel<float, false, true, true>), dim3(grid), dim3(256), 0, s, *f);
        else hipLaunchKernelGGL((mifft::fft_xcd2_kernel<float, false, true, false>), dim3(grid), dim3(256), 0, s, *f);
    }
    return (int)hipGetLastError();
}

// fp32 instances of the XCD-cooperative single-crossing kernel (fft_xcd2.hpp).  -fno-slp-vectorize: see fft_col2_f32.hip.
#include "mifft_internal.h"
#include "fft_xcd2.hpp"

// `mode`: 0 = the kernel; 1..5 = the elimination variants of fft_xcd2.hpp (development; interleaved, prefetching form only)
extern "C" int mifft_xcd2_f32_launch(const mifft::Xcd2Args* f, int split, int prefetch, int mode, unsigned grid, hipStream_t s) {
    if (mode != 0) {
        if (split || !prefetch) return -2;
        switch (mode) {
            case 1: hipLaunchKernelGGL((mifft::fft_xcd2_kernel<float, false, true, true, 1>), dim3(grid), dim3(256), 0, s, *f); break;
            case 2: hipLaunchKernelGGL((mifft::fft_xcd2_kernel<float, false, true, true, 2>), dim3(grid), dim3(256), 0, s, *f); break;
            case 3: hipLaunchKernelGGL((mifft::fft_xcd2_kernel<float, false, true, true, 3>), dim3(grid), dim3(256), 0, s, *f); break;
            case 5: hipLaunchKernelGGL((mifft::fft_xcd2_kernel<float, false, true, true, 5>), dim3(grid), dim3(256), 0, s, *f); break;
            default: return -2;
        }
        return (int)hipGetLastError();
    }
    if (split) {
        if (prefetch) hipLaunchKernelGGL((mifft::fft_xcd2_kernel<float, true, true, true>), dim3(grid), dim3(256), 0, s, *f);
        else hipLaunchKernelGGL((mifft::fft_xcd2_kernel<float, true, true, false>), dim3(grid), dim3(256), 0, s, *f);
    } else {
        if (prefetch) hipLaunchKernelGGL((mifft::fft_xcd2_kernel<float, false, true, true>), dim3(grid), dim3(256), 0, s, *f);
        else hipLaunchKernelGGL((mifft::fft_xcd2_kernel<float, false, true, false>), dim3(grid), dim3(256), 0, s, *f);
    }
    return (int)hipGetLastError();
}

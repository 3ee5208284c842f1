// fp64 instance of the fused persistent two-pass kernel (fft_fused2.hpp) on the 512-thread tiles of fft_col3.hpp:
// N = 2^20 = 1024 x 1024.  -fno-slp-vectorize: see fft_col2_f32.hip.
#include "mifft_internal.h"
#include "fft_fused2.hpp"

extern "C" int mifft_fused3_f64_launch(int L0, int L1, const mifft::FusedArgs* f, int split, unsigned grid, hipStream_t s) {
    if (L0 != 1024 || L1 != 1024) return MIFFT_E_UNSUPPORTED;
    if (split) hipLaunchKernelGGL((mifft::fft_fused3_kernel<double, 2, 2, true, false>), dim3(grid), dim3(512), 0, s, *f);
    else hipLaunchKernelGGL((mifft::fft_fused3_kernel<double, 2, 2, false, true>), dim3(grid), dim3(512), 0, s, *f);
    return (int)hipGetLastError();
}

// 2-D 1024 x 1024 (fft_fused3d_kernel)
extern "C" int mifft_fused3d_f64_launch(int L, const mifft::FusedArgs* f, int split, unsigned grid, hipStream_t s) {
    if (L != 1024) return MIFFT_E_UNSUPPORTED;
    if (split) hipLaunchKernelGGL((mifft::fft_fused3d_kernel<double, 2, 2, true, false>), dim3(grid), dim3(512), 0, s, *f);
    else hipLaunchKernelGGL((mifft::fft_fused3d_kernel<double, 2, 2, false, true>), dim3(grid), dim3(512), 0, s, *f);
    return (int)hipGetLastError();
}

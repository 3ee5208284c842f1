// fp32 instances of the fused 2-D kernels for RECTANGLES (fft_fused2.hpp; round 4): (ny, nx) in {512, 1024, 2048}^2, ny != nx,
// interleaved.  Both sides up to 1024: 256-thread tiles (fft_fused2d_kernel); a 2048-point axis: 512-thread tiles
// (fft_fused3d_kernel, axis length 512 * A).  -fno-slp-vectorize: see fft_col2_f32.hip.
#include "../../include/mifft.h"
#include "mifft_internal.h"
#include "fft_fused2.hpp"

extern "C" int mifft_fused2d_rect_f32_launch(int ny, int nx, const mifft::FusedArgs* f, unsigned grid, hipStream_t s) {
#define RECT2(NY, NX)                                                                                                             \
    if (ny == NY && nx == NX) {                                                                                                   \
        hipLaunchKernelGGL((mifft::fft_fused2d_kernel<float, NY / 256, NX / 256, false, true>), dim3(grid), dim3(256), 0, s, *f); \
        return (int)hipGetLastError();                                                                                            \
    }
#define RECT3(NY, NX)                                                                                                             \
    if (ny == NY && nx == NX) {                                                                                                   \
        hipLaunchKernelGGL((mifft::fft_fused3d_kernel<float, NY / 512, NX / 512, false, true>), dim3(grid), dim3(512), 0, s, *f); \
        return (int)hipGetLastError();                                                                                            \
    }
    RECT2(512, 1024)
    RECT2(1024, 512)
    RECT3(1024, 2048)
    RECT3(2048, 1024)
    RECT3(512, 2048)
    RECT3(2048, 512)
#undef RECT2
#undef RECT3
    return MIFFT_E_UNSUPPORTED;
}

// 2-D shapes with a 256- or 512-point axis on 32-column tiles for that axis' pass (fft_fused2dw_kernel); *tiles0 / *tiles1 per transform
extern "C" int mifft_fused2dw_f32(int ny, int nx, const mifft::FusedArgs* f, unsigned grid, hipStream_t s, int query, unsigned* tiles0,
                                  unsigned* tiles1) {
#define WIDE2D(NY, NX)                                                                                                   \
    if (ny == NY && nx == NX) {                                                                                          \
        if (tiles0) *tiles0 = NX / (NY <= 512 ? 32 : 16);                                                                \
        if (tiles1) *tiles1 = NY / (NX <= 512 ? 32 : 16);                                                                \
        if (query) return 0;                                                                                             \
        hipLaunchKernelGGL((mifft::fft_fused2dw_kernel<NY / 256, NX / 256>), dim3(grid), dim3(256), 0, s, *f);           \
        return (int)hipGetLastError();                                                                                   \
    }
    WIDE2D(512, 512)
    WIDE2D(512, 1024)
    WIDE2D(1024, 512)
    // a 256-point axis (round 4, second batch): 64 KiB tiles of 32 columns, as in the 1-D kernel for N = 2^16
    WIDE2D(256, 256)
    WIDE2D(256, 512)
    WIDE2D(512, 256)
    WIDE2D(256, 1024)
    WIDE2D(1024, 256)
#undef WIDE2D
    return MIFFT_E_UNSUPPORTED;
}

// 1-D N = 2^16 ... 2^18 on the 32-column tiles (fft_fused2w_kernel): L0 >= L1 in {256, 512}, interleaved
extern "C" int mifft_fused2w_f32_launch(int L0, int L1, const mifft::FusedArgs* f, unsigned grid, hipStream_t s) {
#define WIDE(A0, A1)                                                                                            \
    if (L0 == 256 * A0 && L1 == 256 * A1) {                                                                     \
        hipLaunchKernelGGL((mifft::fft_fused2w_kernel<A0, A1>), dim3(grid), dim3(256), 0, s, *f);               \
        return (int)hipGetLastError();                                                                          \
    }
    WIDE(1, 1)
    WIDE(2, 1)
    WIDE(2, 2)
#undef WIDE
    if (L0 == 1024 && L1 == 512) {      // mixed: 16-column first pass, 32-column second pass
        hipLaunchKernelGGL((mifft::fft_fused2m_kernel<1>), dim3(grid), dim3(256), 0, s, *f);
        return (int)hipGetLastError();
    }
    return MIFFT_E_UNSUPPORTED;
}

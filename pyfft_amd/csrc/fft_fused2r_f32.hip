// Instances of the row-first persistent 2-D kernel for split-complex fp32 planes (fft_fused2r.hpp): (ny, nx) in {256, 512, 1024}^2.
// -fno-slp-vectorize: see fft_col2_f32.hip.
#include "../../include/mifft.h"
#include "mifft_internal.h"
#include "fft_fused2r.hpp"

extern "C" int mifft_fused2r_f32(int ny, int nx, const mifft::FusedArgs* f, unsigned grid, hipStream_t s, int query) {
#define RF(NY, NX)                                                                                           \
    if (ny == NY && nx == NX) {                                                                              \
        if (query) return 0;                                                                                 \
        hipLaunchKernelGGL((mifft::fft_fused2r_kernel<NX, NY / 256>), dim3(grid), dim3(256), 0, s, *f);      \
        return (int)hipGetLastError();                                                                       \
    }
    RF(1024, 1024)
    RF(512, 512)
    RF(512, 1024)
    RF(1024, 512)
    RF(256, 256)
    RF(256, 512)
    RF(512, 256)
    RF(256, 1024)
    RF(1024, 256)
#undef RF
    return MIFFT_E_UNSUPPORTED;
}

// fp64 instance of the fused persistent two-pass kernel (fft_fused2.hpp) on the 512-thread tiles of fft_col3.hpp:
// N = 2^20 = 1024 x 1024.  -fno-slp-vectorize: see fft_col2_f32.hip.
#include "mifft_internal.h"
#include "fft_fused2.hpp"

// round 4: N = 2^16 ... 2^18 = L0 x L1 with L0 >= L1 in {256, 512} on the 256-thread two-phase tiles (fft_fused2_kernel<double>: 16-byte
// points cross LDS one component at a time, write-through intermediate by 16-byte sc1 stores), interleaved
namespace {
template <int A0, int A1> int launch2(const mifft::FusedArgs* f, int split, unsigned grid, hipStream_t s) {
    // (split planes, second batch of round 4: 16 columns of an fp64 plane are a whole 128-byte line, streamed non-temporally)
    if (split) hipLaunchKernelGGL((mifft::fft_fused2_kernel<double, A0, A1, true, 1>), dim3(grid), dim3(256), 0, s, *f);
    else hipLaunchKernelGGL((mifft::fft_fused2_kernel<double, A0, A1, false, 1>), dim3(grid), dim3(256), 0, s, *f);
    return (int)hipGetLastError();
}
}  // namespace

extern "C" int mifft_fused3_f64_launch(int L0, int L1, const mifft::FusedArgs* f, int split, unsigned grid, hipStream_t s) {
    if (L0 == 512 && L1 == 512) return launch2<2, 2>(f, split, grid, s);
    if (L0 == 512 && L1 == 256) return launch2<2, 1>(f, split, grid, s);
    if (L0 == 256 && L1 == 256) return launch2<1, 1>(f, split, grid, s);
    if (L0 == 1024 && L1 == 512) {      // 2^19: the 512-point pass on the 512-thread tiles too (2 x 256 by decimation in time)
        if (split) hipLaunchKernelGGL((mifft::fft_fused3_kernel<double, 2, 1, true, true>), dim3(grid), dim3(512), 0, s, *f);
        else hipLaunchKernelGGL((mifft::fft_fused3_kernel<double, 2, 1, false, true>), dim3(grid), dim3(512), 0, s, *f);
        return (int)hipGetLastError();
    }
    if (L0 != 1024 || L1 != 1024) return MIFFT_E_UNSUPPORTED;
    // (split planes: non-temporal loads and stores of the planes since the second batch of round 4 -- fft_col3.hpp honoured neither
    // hint for planes before: 2^20 0.373 -> 0.406, profiles/r04_aj_split_nt_ab.log)
    if (split) hipLaunchKernelGGL((mifft::fft_fused3_kernel<double, 2, 2, true, true>), dim3(grid), dim3(512), 0, s, *f);
    else hipLaunchKernelGGL((mifft::fft_fused3_kernel<double, 2, 2, false, true>), dim3(grid), dim3(512), 0, s, *f);
    return (int)hipGetLastError();
}

// 2-D (ny, nx) in {512, 1024}^2: 1024 x 1024 on the 512-thread tiles (fft_fused3d_kernel, split planes too); round 4: (512, 512) on the
// 256-thread two-phase tiles (fft_fused2d_kernel<double>), (512, 1024) / (1024, 512) on the 512-thread tiles (axis length 512 * A)
extern "C" int mifft_fused3d_f64_launch(int ny, int nx, const mifft::FusedArgs* f, int split, unsigned grid, hipStream_t s) {
#define SMALL2D(NY, NX)                                                                                                            \
    if (!split && ny == NY && nx == NX) {                                                                                          \
        hipLaunchKernelGGL((mifft::fft_fused2d_kernel<double, NY / 256, NX / 256, false, true>), dim3(grid), dim3(256), 0, s, *f); \
        return (int)hipGetLastError();                                                                                             \
    }
    SMALL2D(512, 512)
    SMALL2D(256, 256)      // (a 256-point axis: round 4, second batch)
    SMALL2D(256, 512)
    SMALL2D(512, 256)
#undef SMALL2D
    if (!split && ny == 512 && nx == 1024) {
        hipLaunchKernelGGL((mifft::fft_fused3d_kernel<double, 1, 2, false, true>), dim3(grid), dim3(512), 0, s, *f);
        return (int)hipGetLastError();
    }
    if (!split && ny == 1024 && nx == 512) {
        hipLaunchKernelGGL((mifft::fft_fused3d_kernel<double, 2, 1, false, true>), dim3(grid), dim3(512), 0, s, *f);
        return (int)hipGetLastError();
    }
    if (ny != 1024 || nx != 1024) return MIFFT_E_UNSUPPORTED;
    if (split) hipLaunchKernelGGL((mifft::fft_fused3d_kernel<double, 2, 2, true, true>), dim3(grid), dim3(512), 0, s, *f);   // (1024^2 split: 0.374 -> 0.423)
    else hipLaunchKernelGGL((mifft::fft_fused3d_kernel<double, 2, 2, false, true>), dim3(grid), dim3(512), 0, s, *f);
    return (int)hipGetLastError();
}

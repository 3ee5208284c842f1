// fp32 instances of the fused two-pass kernel (fft_fused2.hpp).  -fno-slp-vectorize: see fft_col2_f32.hip.
#include "../../include/mifft.h"
#include "mifft_internal.h"
#include "fft_fused2.hpp"

namespace {
// The default build of the library holds the forms plans choose: split planes on the lane-interleaved sibling tiles, interleaved data
// with a 1024-point pass on the 16-column tiles (shorter passes run the 32-column tiles of fft_fused2d_f32.hip), both streamed
// non-temporally, L0 >= L1 (the chain's factorisation puts the larger radix first).  `make DEV=1` adds the A/B forms behind the
// development switches: 16-column tiles for every length and one tile per 256-thread work-group for planes (MIFFT_NARROW_TILES=1),
// plain / write-through streams (MIFFT_FUSED_NO_NT, MIFFT_STORE), L0 < L1.  A switch that asks for a form this build lacks gets -2.
template <int A0, int A1> int launch(const mifft::FusedArgs* f, int split, unsigned grid, hipStream_t s) {
    // non-temporal accesses on the streamed side (input of pass 1, output of pass 2) leave the Infinity Cache to the
    // intermediate ring: C2 35.0 -> 36.6 % (development switch to turn it off)
    const bool nt = mifft_debug_get(MIFFT_DEBUG_FUSED_NO_NT) == 0;
    const bool wt = mifft_debug_get(MIFFT_DEBUG_STORE) == 2;   // A/B: write-through stores of the output
    const bool narrow = mifft_debug_get(MIFFT_DEBUG_NARROW_TILES) == 1;
    // (round 4: every interleaved size streams non-temporally, not only 1024 x 1024 -- the XCD-local development kernel always
    // did, which was part of its lead at 2^19; the plain and write-through forms stay as A/B instances of 1024 x 1024)
    // split planes: the sibling 16-column tiles side by side in a 512-thread work-group (fft_fused2s_kernel); MIFFT_NARROW_TILES=1: the
    // round-2 form, one tile per 256-thread work-group (A/B)
#ifdef MIFFT_DEV_BUILD
    if (split && !narrow) {
        // the planes streamed non-temporally, like interleaved data (whole lines per wave instruction: 2^20 0.366 -> 0.420); A/B:
        // MIFFT_STORE = 3 plain accesses
        if constexpr (A0 >= A1) {      // (the shapes plans build)
            if (mifft_debug_get(MIFFT_DEBUG_STORE) == 3) {
                hipLaunchKernelGGL((mifft::fft_fused2s_kernel<A0, A1, false, false>), dim3(grid), dim3(512), 0, s, *f);
                return (int)hipGetLastError();
            }
        }
        hipLaunchKernelGGL((mifft::fft_fused2s_kernel<A0, A1, false, true>), dim3(grid), dim3(512), 0, s, *f);
    }
    else if (split)
        hipLaunchKernelGGL((mifft::fft_fused2_kernel<float, A0, A1, true, 0>), dim3(grid), dim3(256), 0, s, *f);

    else if (nt && wt && A0 == 4 && A1 == 4)
        hipLaunchKernelGGL((mifft::fft_fused2_kernel<float, A0, A1, false, 2>), dim3(grid), dim3(256), 0, s, *f);
    else if (!nt && A0 == 4 && A1 == 4)
        hipLaunchKernelGGL((mifft::fft_fused2_kernel<float, A0, A1, false, 0>), dim3(grid), dim3(256), 0, s, *f);
    else
        hipLaunchKernelGGL((mifft::fft_fused2_kernel<float, A0, A1, false, 1>), dim3(grid), dim3(256), 0, s, *f);
    return (int)hipGetLastError();
#else
    if constexpr (A0 < A1) {
        return -2;
    } else {
        if (narrow || !nt || wt || mifft_debug_get(MIFFT_DEBUG_STORE) == 3) return -2;
        if (split) {
            hipLaunchKernelGGL((mifft::fft_fused2s_kernel<A0, A1, false, true>), dim3(grid), dim3(512), 0, s, *f);
        } else if constexpr (A0 == 4) {
            hipLaunchKernelGGL((mifft::fft_fused2_kernel<float, A0, A1, false, 1>), dim3(grid), dim3(256), 0, s, *f);
        } else {
            return -2;     // (interleaved L0, L1 <= 512: the 32-column tiles, mifft_fused2w_f32_launch)
        }
        return (int)hipGetLastError();
    }
#endif
}
}  // namespace

extern "C" int mifft_fused2_f32_launch(int L0, int L1, const mifft::FusedArgs* f, int split, unsigned grid, hipStream_t s) {
    if (L0 == 2048 && L1 == 2048) {   // 512-thread tiles (fft_col3.hpp)
        if (split) hipLaunchKernelGGL((mifft::fft_fused3_kernel<float, 4, 4, true, true>), dim3(grid), dim3(512), 0, s, *f);   // (planes: non-temporal, 0.302 -> 0.312)
#ifdef MIFFT_DEV_BUILD
        // (round 6, measured and not adopted: the list that issues a tile's loads BEFORE the publish of the previous one, with the item handed
        // out from inside the previous tile -- 0.392 against 0.393 on configuration 5, profiles/r06_d_loads_first_list_ab.log; MIFFT_DEBUG_PREFETCH = 1)
        else if (f->c.lag != 0u && mifft_debug_get(MIFFT_DEBUG_PREFETCH) == 1)
            hipLaunchKernelGGL((mifft::fft_fused3p_kernel<float, 4, 4, true>), dim3(grid), dim3(512), 0, s, *f);
#endif
        else hipLaunchKernelGGL((mifft::fft_fused3_kernel<float, 4, 4, false, true>), dim3(grid), dim3(512), 0, s, *f);
        return (int)hipGetLastError();
    }
    if (L0 == 2048 && L1 == 1024) {
        if (split) hipLaunchKernelGGL((mifft::fft_fused3_kernel<float, 4, 2, true, true>), dim3(grid), dim3(512), 0, s, *f);
#ifdef MIFFT_DEV_BUILD
        else if (f->c.lag != 0u && mifft_debug_get(MIFFT_DEBUG_PREFETCH) == 1)
            hipLaunchKernelGGL((mifft::fft_fused3p_kernel<float, 4, 2, true>), dim3(grid), dim3(512), 0, s, *f);
#endif
        else hipLaunchKernelGGL((mifft::fft_fused3_kernel<float, 4, 2, false, true>), dim3(grid), dim3(512), 0, s, *f);
        return (int)hipGetLastError();
    }
    const int key = (L0 / 256) * 10 + (L1 / 256);
    switch (key) {
        case 11: return launch<1, 1>(f, split, grid, s);
        case 12: return launch<1, 2>(f, split, grid, s);
        case 14: return launch<1, 4>(f, split, grid, s);
        case 21: return launch<2, 1>(f, split, grid, s);
        case 22: return launch<2, 2>(f, split, grid, s);
        case 24: return launch<2, 4>(f, split, grid, s);
        case 41: return launch<4, 1>(f, split, grid, s);
        case 42: return launch<4, 2>(f, split, grid, s);
        case 44: return launch<4, 4>(f, split, grid, s);
    }
    return -2;
}

// XCD-local work lists (strategy `fusedx`): fp32, L0 >= L1 in {256, 512, 1024}; split planes (user side) too
// (a development strategy: instantiated by `make DEV=1` only, mifft_has_feature(MIFFT_FEATURE_FUSED2X))
extern "C" int mifft_fused2x_f32_launch(int L0, int L1, const mifft::FusedArgs* f, int split, unsigned grid, hipStream_t s) {
#ifndef MIFFT_DEV_BUILD
    (void)L0; (void)L1; (void)f; (void)split; (void)grid; (void)s;
    return -2;
#else
#define XL(A0, A1)                                                                                                  \
    if (L0 == 256 * A0 && L1 == 256 * A1) {                                                                         \
        if (split) hipLaunchKernelGGL((mifft::fft_fused2x_kernel<float, A0, A1, true>), dim3(grid), dim3(256), 0, s, *f); \
        else hipLaunchKernelGGL((mifft::fft_fused2x_kernel<float, A0, A1, false>), dim3(grid), dim3(256), 0, s, *f);      \
        return (int)hipGetLastError();                                                                              \
    }
    XL(1, 1)
    XL(2, 1)
    XL(2, 2)
    XL(4, 2)
    XL(4, 4)
#undef XL
    return -2;
#endif
}

// 2-D squares: 512 and 1024 on the 256-thread tiles (fft_fused2d_kernel), 2048 on the 512-thread ones (fft_fused3d_kernel);
// the rectangles live in fft_fused2d_f32.hip
extern "C" int mifft_fused2d_rect_f32_launch(int ny, int nx, const mifft::FusedArgs* f, unsigned grid, hipStream_t s);
extern "C" int mifft_fused2d_f32_launch(int ny, int nx, const mifft::FusedArgs* f, int split, unsigned grid, hipStream_t s) {
    if (ny != nx) return split ? MIFFT_E_UNSUPPORTED : mifft_fused2d_rect_f32_launch(ny, nx, f, grid, s);
    const int L = nx;
#ifndef MIFFT_DEV_BUILD
    // (default build: interleaved squares only -- 1024 on the 16-column tiles, 2048 on the 512-thread ones; 512 runs the 32-column tiles
    // of fft_fused2d_f32.hip.  Split-complex squares run the row-first kernel, fft_fused2r_f32.hip; their two-transposing-pass forms, on
    // request only and slower than the pipelined chunks, and the 16-column form of 512 x 512 are `make DEV=1` instances)
    if (split || (L != 1024 && L != 2048)) return MIFFT_E_UNSUPPORTED;
#endif
#ifdef MIFFT_DEV_BUILD
    const bool sib = split && mifft_debug_get(MIFFT_DEBUG_NARROW_TILES) != 1;   // split planes: sibling tiles side by side
    if (L == 512) {
        if (sib) hipLaunchKernelGGL((mifft::fft_fused2s_kernel<2, 2, true, true>), dim3(grid), dim3(512), 0, s, *f);
        else if (split) hipLaunchKernelGGL((mifft::fft_fused2d_kernel<float, 2, 2, true, false>), dim3(grid), dim3(256), 0, s, *f);
        else hipLaunchKernelGGL((mifft::fft_fused2d_kernel<float, 2, 2, false, true>), dim3(grid), dim3(256), 0, s, *f);
        return (int)hipGetLastError();
    }
#endif
    if (L == 2048) {   // 512-thread tiles
#ifdef MIFFT_DEV_BUILD
        if (split) hipLaunchKernelGGL((mifft::fft_fused3d_kernel<float, 4, 4, true, true>), dim3(grid), dim3(512), 0, s, *f);
        else
#endif
        hipLaunchKernelGGL((mifft::fft_fused3d_kernel<float, 4, 4, false, true>), dim3(grid), dim3(512), 0, s, *f);
        return (int)hipGetLastError();
    }
    if (L != 1024) return MIFFT_E_UNSUPPORTED;
#ifdef MIFFT_DEV_BUILD
    if (sib) hipLaunchKernelGGL((mifft::fft_fused2s_kernel<4, 4, true, true>), dim3(grid), dim3(512), 0, s, *f);
    else if (split) hipLaunchKernelGGL((mifft::fft_fused2d_kernel<float, 4, 4, true, false>), dim3(grid), dim3(256), 0, s, *f);
    else
#endif
    hipLaunchKernelGGL((mifft::fft_fused2d_kernel<float, 4, 4, false, true>), dim3(grid), dim3(256), 0, s, *f);
    return (int)hipGetLastError();
}

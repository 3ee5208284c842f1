// fp32 strided-axis (COL) tile kernels: P = L*W points per work-group, 16 points per thread.
#include "mifft_internal.h"

// two-phase kernels for L = 256/512/1024 live in their own translation unit (fft_col2_f32.hip)
extern "C" int mifft_col2_f32_eligible(int L, int tr, const mifft::TileArgs* a);
extern "C" int mifft_col2_f32_launch(int L, int tr, const mifft::TileArgs* a, hipStream_t s);
// the 512-thread two-phase kernel for L = 2048 (fft_col3_f32.hip)
extern "C" int mifft_col3_f32_eligible(int L, int tr, const mifft::TileArgs* a);
extern "C" int mifft_col3_f32_launch(int tr, const mifft::TileArgs* a, hipStream_t s);

// register-only kernels for L <= 32 in the plain form (fft_colr.hip)
extern "C" int mifft_colr_eligible(int f64, int L, int tr, const mifft::TileArgs* a);
extern "C" int mifft_colr_launch(int f64, int L, const mifft::TileArgs* a, hipStream_t s);

// variant 0: library default (two-phase kernel for L = 256/512/1024 when the tile is 16 whole columns of one
//            matrix, i.e. M*S >= 16; generic tile kernel otherwise);  variant 1: always the generic tile kernel.
extern "C" int mifft_dispatch_col_f32(int L, int tr, int variant, const mifft::TileArgs* a, hipStream_t s, int query_only) {
    if (variant != 0 && variant != 1) return -2;
    if (variant == 0 && !query_only && mifft_colr_eligible(0, L, tr, a)) return mifft_colr_launch(0, L, a, s);
    if (variant == 0 && !query_only && mifft_col2_f32_eligible(L, tr, a)) return mifft_col2_f32_launch(L, tr, a, s);
    if (variant == 0 && !query_only && mifft_col3_f32_eligible(L, tr, a)) return mifft_col3_f32_launch(tr, a, s);
    switch (L) {
        MIFFT_COL_CASE(float, 2, 2048, 256, 2)
        MIFFT_COL_CASE(float, 4, 1024, 256, 4)
        MIFFT_COL_CASE(float, 8, 512, 256, 8)
        MIFFT_COL_CASE(float, 16, 256, 256, 16)
        MIFFT_COL_CASE(float, 32, 128, 256, 8, 4)
        MIFFT_COL_CASE(float, 64, 64, 256, 8, 8)
        MIFFT_COL_CASE(float, 128, 32, 256, 16, 8)
        MIFFT_COL_CASE(float, 256, 16, 256, 16, 16)
        MIFFT_COL_CASE(float, 512, 16, 512, 8, 8, 8)
        MIFFT_COL_CASE(float, 1024, 16, 1024, 16, 16, 4)
        MIFFT_COL_CASE(float, 2048, 8, 1024, 16, 16, 8)   // (fallback of the L = 2048 two-phase kernel: 128 KiB tile of 8 columns)
    }
    return -2;
}

// fp64 strided-axis (COL) tile kernels: 8 points per thread, W >= 8 columns (128-byte segments).
#include "mifft_internal.h"
extern "C" int mifft_col2_f64_eligible(int L, int tr, const mifft::TileArgs* a);
extern "C" int mifft_col2_f64_launch(int L, int tr, const mifft::TileArgs* a, hipStream_t s);
// the 512-thread two-phase kernel for L = 1024 (fft_col3_f64.hip)
extern "C" int mifft_col3_f64_eligible(int L, int tr, const mifft::TileArgs* a);
extern "C" int mifft_col3_f64_launch(int tr, const mifft::TileArgs* a, hipStream_t s);

// the stage-chain kernel for L = 2048 (fft_colx_f64.hip)
extern "C" int mifft_colx_f64_eligible(int L, int tr, const mifft::TileArgs* a);
extern "C" int mifft_colx_f64_launch(int tr, const mifft::TileArgs* a, hipStream_t s);

// register-only kernels for L <= 32 in the plain form (fft_colr.hip)
extern "C" int mifft_colr_eligible(int f64, int L, int tr, const mifft::TileArgs* a);
extern "C" int mifft_colr_launch(int f64, int L, const mifft::TileArgs* a, hipStream_t s);

// variant 0: library default (two-phase kernel for L = 256 when eligible); variant 1: always the generic tile kernel
extern "C" int mifft_dispatch_col_f64(int L, int tr, int variant, const mifft::TileArgs* a, hipStream_t s, int query_only) {
    if (variant != 0 && variant != 1) return -2;
    if (variant == 0 && !query_only && mifft_colr_eligible(1, L, tr, a)) return mifft_colr_launch(1, L, a, s);
    if (variant == 0 && !query_only && mifft_col2_f64_eligible(L, tr, a)) return mifft_col2_f64_launch(L, tr, a, s);
    if (variant == 0 && !query_only && mifft_col3_f64_eligible(L, tr, a)) return mifft_col3_f64_launch(tr, a, s);
    if (variant == 0 && !query_only && mifft_colx_f64_eligible(L, tr, a)) return mifft_colx_f64_launch(tr, a, s);
    switch (L) {
        MIFFT_COL_CASE(double, 2, 1024, 256, 2)
        MIFFT_COL_CASE(double, 4, 512, 256, 4)
        MIFFT_COL_CASE(double, 8, 256, 256, 8)
        MIFFT_COL_CASE(double, 16, 128, 256, 4, 4)
        MIFFT_COL_CASE(double, 32, 64, 256, 8, 4)
        MIFFT_COL_CASE(double, 64, 32, 256, 8, 8)
        MIFFT_COL_CASE(double, 128, 16, 256, 8, 4, 4)
        MIFFT_COL_CASE(double, 256, 8, 256, 8, 8, 4)
        MIFFT_COL_CASE(double, 512, 8, 512, 8, 8, 8)
        MIFFT_COL_CASE(double, 1024, 8, 1024, 8, 8, 4, 4)
        MIFFT_COL_CASE(double, 2048, 4, 1024, 8, 8, 8, 4)   // (fallback of the stage-chain kernel: 128 KiB tile of 4 columns)
    }
    return -2;
}

// fp32 strided-axis (COL) tile kernels: P = L*W points per work-group, 16 points per thread.
#include "mifft_internal.h"
extern "C" int mifft_dispatch_col_f32(int L, int tr, int variant, const mifft::TileArgs* a, hipStream_t s, int query_only) {
    if (variant != 0) return -2;
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
    }
    return -2;
}

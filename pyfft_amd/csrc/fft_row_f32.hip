// fp32 contiguous-axis (ROW) tile kernels: W rows of L points per work-group, 16 points per thread.
#include "mifft_internal.h"
extern "C" int mifft_dispatch_row_f32(int L, int variant, const mifft::TileArgs* a, hipStream_t s, int query_only) {
    if (variant != 0) return -2;
    switch (L) {
        MIFFT_ROW_CASE(float, 2, 2048, 256, 2)
        MIFFT_ROW_CASE(float, 4, 1024, 256, 4)
        MIFFT_ROW_CASE(float, 8, 512, 256, 8)
        MIFFT_ROW_CASE(float, 16, 256, 256, 16)
        MIFFT_ROW_CASE(float, 32, 128, 256, 8, 4)
        MIFFT_ROW_CASE(float, 64, 64, 256, 8, 8)
        MIFFT_ROW_CASE(float, 128, 32, 256, 16, 8)
        MIFFT_ROW_CASE(float, 256, 16, 256, 16, 16)
        MIFFT_ROW_CASE(float, 512, 8, 256, 8, 8, 8)
        MIFFT_ROW_CASE(float, 1024, 4, 256, 16, 16, 4)
        MIFFT_ROW_CASE(float, 2048, 2, 256, 16, 16, 8)
        MIFFT_ROW_CASE(float, 4096, 1, 256, 16, 16, 16)
        MIFFT_ROW_CASE(float, 8192, 1, 512, 16, 16, 16, 2)
        MIFFT_ROW_CASE(float, 16384, 1, 1024, 16, 16, 16, 4)
    }
    return -2;
}

// fp64 contiguous-axis (ROW) tile kernels: 8 points per thread.
#include "mifft_internal.h"
extern "C" int mifft_dispatch_row_f64(int L, int variant, const mifft::TileArgs* a, hipStream_t s, int query_only) {
    if (variant != 0) return -2;
    switch (L) {
        MIFFT_ROW_CASE(double, 2, 1024, 256, 2)
        MIFFT_ROW_CASE(double, 4, 512, 256, 4)
        MIFFT_ROW_CASE(double, 8, 256, 256, 8)
        MIFFT_ROW_CASE(double, 16, 128, 256, 4, 4)
        MIFFT_ROW_CASE(double, 32, 64, 256, 8, 4)
        MIFFT_ROW_CASE(double, 64, 32, 256, 8, 8)
        MIFFT_ROW_CASE(double, 128, 16, 256, 8, 4, 4)
        MIFFT_ROW_CASE(double, 256, 8, 256, 8, 8, 4)
        MIFFT_ROW_CASE(double, 512, 4, 256, 8, 8, 8)
        MIFFT_ROW_CASE(double, 1024, 2, 256, 8, 8, 4, 4)
        MIFFT_ROW_CASE(double, 2048, 1, 256, 8, 8, 8, 4)
        MIFFT_ROW_CASE(double, 4096, 1, 512, 8, 8, 8, 8)
        MIFFT_ROW_CASE(double, 8192, 1, 1024, 8, 8, 8, 8, 2)
    }
    return -2;
}

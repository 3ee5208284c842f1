// fp64 contiguous-axis (ROW) kernels: register-edged form (fft_row2.hpp) for interleaved L >= 1024, LDS-staged tile
// kernels (8 points per thread) otherwise.
#include "mifft_internal.h"
#include "fft_row2.hpp"
extern "C" int mifft_dispatch_row_f64(int L, int variant, const mifft::TileArgs* a, hipStream_t s, int query_only) {
    if (variant != 0 && variant != 2) return -2;
    // L = 16384 exists in the register-edged half-exchange form only (interleaved on both sides; query with variant 2, MIFFT_VARIANT_INTERLEAVED_ONLY)
    // (second batch of round 4: planes too)
    if (L == 16384) {
        if (query_only) return 0;
        if (!a || (!a->split && a->split_out)) return -2;
        if (a->split) return mifft::launch_row2_lay<double, 16384, 1, 1024, mifft::RadixList<4, 16, 16, 16>, true, 4>(a, s, 0);
        return mifft::launch_row2<double, 16384, 1, 1024, mifft::RadixList<4, 16, 16, 16>, true, 4>(a, s, 0);
    }
    // both sides interleaved: register-edged kernels (fft_row2.hpp); L <= 512: the LDS-staged tile kernels below
    // measure faster for 16-byte points.  8192: half-exchange form, 2 work-groups per CU instead of 1 (59 % -> 70 %).
    // split-complex planes: the same kernels with plane loads / stores (second batch of round 4, see fft_row_f32.hip)
    if (a && !(!a->split && a->split_out) && (a->split || a->split_out) && mifft_debug_get(MIFFT_DEBUG_NARROW_TILES) != 1) {
        using namespace mifft;
        if (L == 1024) return launch_row2_lay<double, 1024, 4, 256, RadixList<16, 4, 16>>(a, s, query_only);
        if (L == 2048) return launch_row2_lay<double, 2048, 1, 128, RadixList<16, 8, 16>>(a, s, query_only);
        if (L == 4096) return launch_row2_lay<double, 4096, 1, 256, RadixList<16, 16, 16>>(a, s, query_only);
        if (L == 8192) return launch_row2_lay<double, 8192, 1, 512, RadixList<2, 16, 16, 16>, true>(a, s, query_only);
    }
    if (a && !a->split && !a->split_out) {
        using namespace mifft;
        if (L == 1024) return launch_row2<double, 1024, 4, 256, RadixList<16, 4, 16>>(a, s, query_only);
        if (L == 2048) return launch_row2<double, 2048, 1, 128, RadixList<16, 8, 16>>(a, s, query_only);
        if (L == 4096) return launch_row2<double, 4096, 1, 256, RadixList<16, 16, 16>>(a, s, query_only);
        if (L == 8192) return launch_row2<double, 8192, 1, 512, RadixList<2, 16, 16, 16>, true>(a, s, query_only);
    }
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

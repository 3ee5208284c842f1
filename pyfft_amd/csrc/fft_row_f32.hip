// fp32 contiguous-axis (ROW) kernels: register-edged form (fft_row2.hpp) for interleaved L >= 256, LDS-staged tile
// kernels (fft_tile.hpp; W rows of L points per work-group, 16 points per thread) for short rows and split planes.
#include "mifft_internal.h"
#include "fft_row2.hpp"
#include "../../include/mifft.h"
extern "C" int mifft_dispatch_row_f32(int L, int variant, const mifft::TileArgs* a, hipStream_t s, int query_only) {
    if (variant != 0 && variant != 2) return -2;
    // L = 32768 exists in the register-edged half-exchange form only (128 KiB of LDS as scalars, one work-group per CU):
    // interleaved data on both sides; a query with variant 2 (MIFFT_VARIANT_INTERLEAVED_ONLY) asks for exactly that
    // (second batch of round 4: planes too -- the first stage loads and the last stage stores either layout, the exchanges are the same)
    if (L == 32768) {
        if (query_only) return 0;
        if (!a || (!a->split && a->split_out)) return -2;
        if (a->split) return mifft::launch_row2_lay<float, 32768, 1, 1024, mifft::RadixList<32, 32, 32>, true, 4>(a, s, 0);
#ifdef MIFFT_DEV_BUILD      // (A/B instances: `make DEV=1`, MIFFT_FEATURE_AB_FORMS)
        if (mifft_debug_get(MIFFT_DEBUG_PERSIST))   // development: persistent + prefetching form (fft_row2.hpp)
            return mifft::launch_row2p<float, 32768, 1024, mifft::RadixList<8, 16, 16, 16>, true, 4>(a, s, 1);
        switch (mifft_debug_get(MIFFT_DEBUG_ALT_ROWS)) {   // development: A/B of the stage lists
            case 1: return mifft::launch_row2<float, 32768, 1, 1024, mifft::RadixList<8, 16, 16, 16>, true, 4>(a, s, 0);
            case 2: return mifft::launch_row2<float, 32768, 1, 512, mifft::RadixList<32, 32, 32>, true, 2>(a, s, 0);
        }
#else
        if (mifft_debug_get(MIFFT_DEBUG_PERSIST) || mifft_debug_get(MIFFT_DEBUG_ALT_ROWS)) return -2;
#endif
        return mifft::launch_row2<float, 32768, 1, 1024, mifft::RadixList<32, 32, 32>, true, 4>(a, s, 0);
    }
    // both sides interleaved: register-edged kernels (fft_row2.hpp).  Shapes chosen by measurement (1 GiB buffers,
    // tools/row_probe.py): the half-exchange form wins where it raises the work-groups per CU (8192: 2 -> 3,
    // 16384: 1 -> 2), the plain form everywhere else.
    // Split-complex planes (second batch of round 4): the same kernels with the first-stage operands loaded from / the last-stage
    // results stored to the two planes -- planes -> planes (a single-pass plan) and planes -> interleaved (the first pass of a
    // multi-pass plan); rounds 1-3 sent every split row to the LDS-staged tile kernels below (1024 x 4096 planes, the reference's
    // 32 MiB protocol: 0.515 against 0.809 interleaved; 8192: 0.418 / 0.645 -- profiles/r04_at_rows_split.log)
    if (a && !(!a->split && a->split_out) && (a->split || a->split_out) && mifft_debug_get(MIFFT_DEBUG_NARROW_TILES) != 1) {
        using namespace mifft;
        // (L = 256: the LDS-staged kernel measured faster for planes, 0.740 against 0.716)
        if (L == 512) return launch_row2_lay<float, 512, 8, 256, RadixList<16, 2, 16>>(a, s, query_only);
        if (L == 1024) return launch_row2_lay<float, 1024, 4, 256, RadixList<16, 4, 16>>(a, s, query_only);
        if (L == 2048) return launch_row2_lay<float, 2048, 1, 128, RadixList<16, 8, 16>>(a, s, query_only);
        if (L == 4096) return launch_row2_lay<float, 4096, 1, 256, RadixList<16, 16, 16>>(a, s, query_only);
        if (L == 8192) return launch_row2_lay<float, 8192, 1, 256, RadixList<16, 16, 32>, true>(a, s, query_only);
        if (L == 16384) return launch_row2_lay<float, 16384, 1, 512, RadixList<4, 16, 16, 16>, true, 4>(a, s, query_only);
    }
    if (a && !a->split && !a->split_out) {
        using namespace mifft;
        if (L == 256) return launch_row2<float, 256, 8, 256, RadixList<8, 8, 4>>(a, s, query_only);
        if (L == 512) return launch_row2<float, 512, 8, 256, RadixList<16, 2, 16>>(a, s, query_only);
        if (L == 1024) return launch_row2<float, 1024, 4, 256, RadixList<16, 4, 16>>(a, s, query_only);
        if (L == 2048) return launch_row2<float, 2048, 1, 128, RadixList<16, 8, 16>>(a, s, query_only);
        if (L == 4096) return launch_row2<float, 4096, 1, 256, RadixList<16, 16, 16>>(a, s, query_only);
        // development: persistent form of the long rows (the next row's loads are issued from the last stage of the current
        // one).  Measured at 1 GiB buffers (tools/row_probe.py): 4096 71.4 -> 73.5 %, 8192 66.1 -> 64.0 %, 16384 63.8 -> 56.3 %,
        // 32768 47.1 -> 47.2 %: the rows that fill a CU are bound by their LDS exchanges, not by exposed load latency
#ifdef MIFFT_DEV_BUILD
        if (!query_only && mifft_debug_get(MIFFT_DEBUG_PERSIST)) {
            if (L == 4096) return launch_row2p<float, 4096, 256, RadixList<16, 16, 16>, false, 1>(a, s, 4);
            if (L == 8192) return launch_row2p<float, 8192, 256, RadixList<16, 16, 32>, true, 1>(a, s, 3);
            if (L == 16384) return launch_row2p<float, 16384, 512, RadixList<4, 16, 16, 16>, true, 4>(a, s, 2);
        }
#endif
        if (L == 8192) return launch_row2<float, 8192, 1, 256, RadixList<16, 16, 32>, true>(a, s, query_only);
#ifdef MIFFT_DEV_BUILD
        if (L == 16384 && !query_only) {
            switch (mifft_debug_get(MIFFT_DEBUG_ALT_ROWS)) {   // development: A/B of the stage lists
                case 2: return launch_row2<float, 16384, 1, 512, RadixList<16, 32, 32>, true, 4>(a, s, 0);
                case 3: return launch_row2<float, 16384, 1, 512, RadixList<32, 16, 32>, true, 4>(a, s, 0);
            }
        }
#endif
        if (L == 16384) return launch_row2<float, 16384, 1, 512, RadixList<4, 16, 16, 16>, true, 4>(a, s, query_only);
    }
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

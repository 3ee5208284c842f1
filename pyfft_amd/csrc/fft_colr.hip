// Instances of the register-only short strided pass (fft_colr.hpp): fp32 L = 4, 8, 16 on two adjacent columns per thread and
// L = 32 on one; fp64 L = 4, 8, 16, 32 on one column per thread.
#include "mifft_internal.h"
#include "fft_colr.hpp"

using namespace mifft;

static int vec_of(int f64, int L) { return (f64 || L == 32) ? 1 : 2; }

// plain form, interleaved both sides, the thread's V columns inside one l (S >= V) and one matrix, nothing ragged
extern "C" int mifft_colr_eligible(int f64, int L, int tr, const TileArgs* a) {
    if (tr || a->split || a->split_out) return 0;
    if (!(L == 4 || L == 8 || L == 16 || L == 32)) return 0;
    const int v = vec_of(f64, L);
    if (a->total <= 0 || (a->total % v) || (1ll << a->logS) < v) return 0;
    return 1;
}

extern "C" int mifft_colr_launch(int f64, int L, const TileArgs* a, hipStream_t s) {
    if (f64) {
        switch (L) {
            case 4: return launch_colr<double, 4, 1>(a, s);
            case 8: return launch_colr<double, 8, 1>(a, s);
            case 16: return launch_colr<double, 16, 1>(a, s);
            case 32: return launch_colr<double, 32, 1>(a, s);
        }
    } else {
        switch (L) {
            case 4: return launch_colr<float, 4, 2>(a, s);
            case 8: return launch_colr<float, 8, 2>(a, s);
            case 16: return launch_colr<float, 16, 2>(a, s);
            case 32: return launch_colr<float, 32, 1>(a, s);
        }
    }
    return -2;
}

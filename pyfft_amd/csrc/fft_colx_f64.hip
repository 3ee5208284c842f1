// fp64 instances of the stage-chain strided pass (fft_colx.hpp): L = 2048 on 8-column tiles.
#include "mifft_internal.h"
#include "fft_colx.hpp"

using namespace mifft;

// 8 whole columns of one matrix, interleaved on both sides, 32-bit per-thread byte offsets ((2047 << logMS) * 16 bytes in,
// (2047 << logS) * 16 out); TR: S == 1 with the inter-pass twiddle, plain: S >= 8 (with or without twiddle)
extern "C" int mifft_colx_f64_eligible(int L, int tr, const TileArgs* a) {
    if (L != 2048 || a->split || a->split_out) return 0;
    if (a->total <= 0 || (a->total & 7) || a->logMS < 3 || a->logMS > 17) return 0;
    if (tr) return a->logS == 0 && a->has_tw != 0;
    return a->logS >= 3 && a->logS <= 17;
}

extern "C" int mifft_colx_f64_launch(int tr, const TileArgs* a, hipStream_t s) {
    const long long tiles = a->total / 8;
    if (tiles > 2147483647ll) return -1;
    const dim3 g((unsigned)tiles), b(1024);
    using RL = RadixList<16, 16, 8>;
    if (tr) hipLaunchKernelGGL((fft_colx_kernel<double, 2048, 8, 1024, true, 4, true, true, RL>), g, b, 0, s, *a);
    else if (a->has_tw) hipLaunchKernelGGL((fft_colx_kernel<double, 2048, 8, 1024, true, 4, false, true, RL>), g, b, 0, s, *a);
    else hipLaunchKernelGGL((fft_colx_kernel<double, 2048, 8, 1024, true, 4, false, false, RL>), g, b, 0, s, *a);
    return (int)hipGetLastError();
}

// fp64 instances of the persistent two-pass kernel on the stage-chain strided tiles of fft_colx.hpp (round 4): N = 2^22 =
// 2048 x 2048 and N = 2^21 = 2048 x 1024 in double precision, which rounds 2-3 ran as two launches per cache-sized chunk
// (0.27 / 0.29 of the roofline).  Work list, ring and hand-off as in fft_fused2.hpp; a work-group is 1024 threads (one per CU):
//   pass 0  colx tile, L = 2048, 8 columns (256 KiB of points), transposing, inter-pass twiddle, write-through into the ring
//   pass 1  2^22: the same tile in the plain strided form; 2^21: L = 1024 on 16 columns (the same 16384 points, 256-byte segments)
// A transform is 64 / 32 MiB, so the ring holds 3 / 7 of them.  Interleaved data only (pyfft/kernel.py:259-283: the chain of a long axis).
#include "mifft_internal.h"
#include "fft_colx.hpp"
#include "fft_fused2.hpp"

using namespace mifft;

namespace {
template <int L1, int W1, typename RL1>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4))) fft_fusedx64_kernel(const FusedArgs f) {
    constexpr int P = 2048 * 8;
    static_assert(L1 * W1 == P, "both tile kinds hold 16384 points");
    __shared__ __attribute__((aligned(16))) double lds[P + P / 16];
    __shared__ unsigned s_item;
    using RL0 = RadixList<16, 16, 8>;
    constexpr unsigned t0 = (unsigned)L1 / 8u, t1 = 2048u / (unsigned)W1;     // tiles per transform: pass 0 over L1 columns, pass 1 over 2048
    constexpr unsigned per0 = t0 >= t1 ? t0 / t1 : 1u, per1 = t1 > t0 ? t1 / t0 : 1u;
    fused_loop<per0, per1, false>(
        f.c, &s_item,
        [&](unsigned t, unsigned slot, unsigned tile, auto) {
            colx_tile<double, 2048, 8, 1024, true, true, true, true, RL0>(f.p0, (long long)t, (long long)slot, (long long)tile * 8, lds, 1);
        },
        [&](unsigned slot, unsigned t, unsigned tile, auto) {
            colx_tile<double, L1, W1, 1024, true, false, false, false, RL1>(f.p1, (long long)slot, (long long)t, (long long)tile * W1, lds, 2);
        });
}
}  // namespace

// tiles0 / tiles1 per transform; query != 0: nothing is launched
extern "C" int mifft_fusedx_f64(int L0, int L1, const FusedArgs* f, unsigned grid, hipStream_t s, int query, unsigned* tiles0, unsigned* tiles1) {
    if (L0 != 2048 || (L1 != 2048 && L1 != 1024)) return MIFFT_E_UNSUPPORTED;
    if (tiles0) *tiles0 = (unsigned)L1 / 8u;
    if (tiles1) *tiles1 = L1 == 2048 ? 256u : 128u;
    if (query) return 0;
    if (L1 == 2048) hipLaunchKernelGGL((fft_fusedx64_kernel<2048, 8, RadixList<16, 16, 8>>), dim3(grid), dim3(1024), 0, s, *f);
    else hipLaunchKernelGGL((fft_fusedx64_kernel<1024, 16, RadixList<16, 8, 8>>), dim3(grid), dim3(1024), 0, s, *f);
    return (int)hipGetLastError();
}

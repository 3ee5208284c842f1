// fp32 two-phase strided-axis kernels (fft_col2.hpp) for L = 256 / 512 / 1024.
// Built with -fno-slp-vectorize: the SLP vectoriser turns the complex multiplies into v_pk_fma_f32 pairs that
// use only half of each 64-bit result, which doubles the live registers of the 64-point-per-thread kernel and
// makes it spill (measured: 228 bytes/lane of scratch with SLP, 36 without).
#include "mifft_internal.h"
#include "fft_col2.hpp"
#include "fft_col2w.hpp"

namespace {
template <int A, bool TR, bool TW> int launch_l(const mifft::TileArgs* a, hipStream_t s) {
    const long long tiles = a->total / 16;
    if (tiles > 2147483647ll) return -1;
    const dim3 g((unsigned)tiles), b(256);
    if (a->split && a->split_out)
        hipLaunchKernelGGL((mifft::fft_col2_kernel<float, A, TR, TW, true, true>), g, b, 0, s, *a);
    else if (a->split)
        hipLaunchKernelGGL((mifft::fft_col2_kernel<float, A, TR, TW, true, false>), g, b, 0, s, *a);
    else if (a->split_out)
        hipLaunchKernelGGL((mifft::fft_col2_kernel<float, A, TR, TW, false, true>), g, b, 0, s, *a);
    else
        hipLaunchKernelGGL((mifft::fft_col2_kernel<float, A, TR, TW, false, false>), g, b, 0, s, *a);
    return (int)hipGetLastError();
}
// Instances: the transposing first pass of a long axis (S == 1, always with the inter-pass twiddle) and the plain
// strided pass (M == 1, no twiddle).  Other combinations use the generic tile kernel.
template <int A> int launch(int tr, const mifft::TileArgs* a, hipStream_t s) {
    return tr ? launch_l<A, true, true>(a, s) : launch_l<A, false, false>(a, s);
}
}  // namespace

// The tile must be 16 whole columns of one matrix (M*S >= 16) and the per-thread 32-bit byte offsets must fit.
extern "C" int mifft_col2_f32_eligible(int L, int tr, const mifft::TileArgs* a) {
    if (L != 256 && L != 512 && L != 1024) return 0;
    if (a->total <= 0 || a->logMS < 4 || a->logMS > 24 || a->logS > 22) return 0;
    return tr ? (a->has_tw != 0) : (a->has_tw == 0);
}

// Round 4: 32-column tiles (fft_col2w.hpp: 16-byte lanes, 256-byte segments) for L = 256 / 512 on interleaved data -- the tile must
// be 32 whole columns of one matrix and the column pair adjacent in the output (S >= 2 in the plain form).  In the PERSISTENT kernel
// they are worth 3-6 points (fft_fused2w_kernel); as plain launches of the chain / the pipelined chunks they measure within +- 2 %
// of the 16-column tiles, in both directions (profiles/r04_r_plain_wide_tiles_ab.log: 2^16 x 512 0.395 / 0.399, 2^18 x 64
// 0.336 / 0.357, (512, 512) x 128 0.372 / 0.351, pipelined 2^16 0.405 / 0.397), so the plain launches keep the 16 columns; the
// instances stay for the A/B: MIFFT_DEBUG_NARROW_TILES = 2 runs them wherever they fit.
//
// Round 6: ONE exception -- rows that lie far apart.  The plain strided pass along the z axis of a 3-D transform of hundreds of MiB (or
// the last pass of N = 2^24) gathers its 256 / 512 rows 512 KiB ... 2 MiB apart; 128-byte segments at that distance stream at 0.55 of
// the roofline, 256-byte segments at 0.61-0.63 (the copy kernels of tools/membench5.hip say the same: 4.37 against 4.93 TB/s).
// (512, 512, 512) 0.249 -> 0.273, (256, 512, 512) 0.250 -> 0.274 (profiles/r06_i_plane_fused_probe.log).  From S = 2^16 points on.
namespace {
template <int A> int launch_w_plain(const mifft::TileArgs* a, hipStream_t s) {
    const long long tiles = a->total / 32;
    if (tiles > 2147483647ll) return -1;
    hipLaunchKernelGGL((mifft::fft_col2w_kernel<A, false, false>), dim3((unsigned)tiles), dim3(256), 0, s, *a);
    return (int)hipGetLastError();
}
bool wide_far_ok(int L, int tr, const mifft::TileArgs* a) {
    const int sw = mifft_debug_get(MIFFT_DEBUG_NARROW_TILES);      // 1: the rounds 1-3 forms everywhere, 3: this exception off (A/B)
    return !tr && (L == 256 || L == 512) && !a->split && !a->split_out && sw != 1 && sw != 3 &&
           (a->total & 31) == 0 && a->logMS >= 5 && a->logS >= 16;
}
#ifdef MIFFT_DEV_BUILD      // (A/B instances: `make DEV=1`, MIFFT_FEATURE_AB_FORMS)
template <int A> int launch_w(int tr, const mifft::TileArgs* a, hipStream_t s) {
    const long long tiles = a->total / 32;
    if (tiles > 2147483647ll) return -1;
    const dim3 g((unsigned)tiles), b(256);
    if (tr) hipLaunchKernelGGL((mifft::fft_col2w_kernel<A, true, true>), g, b, 0, s, *a);
    else hipLaunchKernelGGL((mifft::fft_col2w_kernel<A, false, false>), g, b, 0, s, *a);
    return (int)hipGetLastError();
}
bool wide_ok(int L, int tr, const mifft::TileArgs* a) {
    if (L != 256 && L != 512) return false;
    if (a->split || a->split_out || mifft_debug_get(MIFFT_DEBUG_NARROW_TILES) != 2) return false;
    if ((a->total & 31) || a->logMS < 5 || (!tr && a->logS < 1)) return false;
    return a->total / 32 >= 1024;
}
#endif
}  // namespace

extern "C" int mifft_col2x_f32_eligible(int L, int tr, const mifft::TileArgs* a);   // fft_col2x_f32.hip
extern "C" int mifft_col2x_f32_launch(int L, int tr, const mifft::TileArgs* a, hipStream_t s);

extern "C" int mifft_col2_f32_launch(int L, int tr, const mifft::TileArgs* a, hipStream_t s) {
    if (mifft_col2x_f32_eligible(L, tr, a)) return mifft_col2x_f32_launch(L, tr, a, s);   // split planes: whole lines per wave instruction
#ifdef MIFFT_DEV_BUILD
    if (wide_ok(L, tr, a)) return L == 512 ? launch_w<2>(tr, a, s) : launch_w<1>(tr, a, s);
#endif
    if (wide_far_ok(L, tr, a)) return L == 512 ? launch_w_plain<2>(a, s) : launch_w_plain<1>(a, s);
    switch (L) {
        case 1024: return launch<4>(tr, a, s);
        case 512: return launch<2>(tr, a, s);
        case 256: return launch<1>(tr, a, s);
    }
    return -2;
}

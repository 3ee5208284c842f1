// Internal glue between the C-ABI runtime (mifft_runtime.cpp) and the per-precision kernel tables.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/mifft.h"
#include "fft_tile.hpp"
#include "fft_fused2.hpp"
#include "fft_xcd2.hpp"
#include "fft_nd.hpp"
#include "fft_wave.hpp"
#include "fft_pair.hpp"
#include "fft_nd2t.hpp"
#include "fft_fusedp.hpp"

// Each returns 0 on success, MIFFT_E_UNSUPPORTED (-2) when no kernel is compiled for (L, tr, variant),
// or a hipError_t.  With query_only != 0 nothing is launched.
extern "C" {
int mifft_dispatch_col_f32(int L, int tr, int variant, const mifft::TileArgs* a, hipStream_t s, int query_only);
int mifft_dispatch_row_f32(int L, int variant, const mifft::TileArgs* a, hipStream_t s, int query_only);
int mifft_dispatch_col_f64(int L, int tr, int variant, const mifft::TileArgs* a, hipStream_t s, int query_only);
int mifft_dispatch_row_f64(int L, int variant, const mifft::TileArgs* a, hipStream_t s, int query_only);
int mifft_nd_max_points(int f64);
int mifft_nd_launch(int f64, long long n, const mifft::NdArgs* a, hipStream_t s);
int mifft_nd2_f32_supported(int x, int y, int z);
int mifft_nd2_f32_launch(int x, int y, int z, const mifft::TileArgs* a, hipStream_t s);
int mifft_nd2_f64_supported(int x, int y, int z);
int mifft_nd2_f64_launch(int x, int y, int z, const mifft::TileArgs* a, hipStream_t s);
int mifft_fused2_f32_launch(int L0, int L1, const mifft::FusedArgs* f, int split, unsigned grid, hipStream_t s);
int mifft_fused2dw_f32(int ny, int nx, const mifft::FusedArgs* f, unsigned grid, hipStream_t s, int query, unsigned* tiles0, unsigned* tiles1);
int mifft_fused2r_f32(int ny, int nx, const mifft::FusedArgs* f, unsigned grid, hipStream_t s, int query);
int mifft_fused2w_f32_launch(int L0, int L1, const mifft::FusedArgs* f, unsigned grid, hipStream_t s);
int mifft_fused2x_f32_launch(int L0, int L1, const mifft::FusedArgs* f, int split, unsigned grid, hipStream_t s);
int mifft_fused2d_f32_launch(int ny, int nx, const mifft::FusedArgs* f, int split, unsigned grid, hipStream_t s);
int mifft_fused3d_f64_launch(int ny, int nx, const mifft::FusedArgs* f, int split, unsigned grid, hipStream_t s);
int mifft_fusedx_f64(int L0, int L1, const mifft::FusedArgs* f, unsigned grid, hipStream_t s, int query, unsigned* tiles0, unsigned* tiles1);
int mifft_fused3_f64_launch(int L0, int L1, const mifft::FusedArgs* f, int split, unsigned grid, hipStream_t s);
int mifft_aux_copy_launch(const struct mifft_copy* c, const void* s0, const void* s1, void* d0, void* d1, hipStream_t s);
int mifft_aux_mul_rows_launch(int f64, void* a, const void* b, long long rows, long long n, hipStream_t s);
int mifft_aux_mismatch_launch(const void* a, const void* b, unsigned long long words, unsigned long long* count, hipStream_t s);
int mifft_aux_zero_launch(void* p, unsigned long long nbytes, hipStream_t s);     // nbytes a multiple of 16, p 16-byte aligned
int mifft_wave_supported(int f64, int N);
int mifft_wave_launch(int f64, int N, const mifft::WaveArgs* a, int max_blocks, hipStream_t s);
int mifft_wave_16x16_launch(const mifft::WaveArgs* a, int max_blocks, hipStream_t s);
int mifft_pair_f64(int kind, int k0, int k1, int k2, int split, const mifft::PairArgs* a, hipStream_t s, int query, int* width);
int mifft_pair_f32(int kind, int k0, int k1, int k2, int split, const mifft::PairArgs* a, hipStream_t s, int query, int* width);
int mifft_fusedp(int f64, int split, int x, int y, int z, const mifft::FusedPairArgs* f, unsigned grid, hipStream_t s, int query, int* r0,
                 unsigned* tiles0, unsigned* tiles1);
int mifft_fusedp_more(int f64, int split, int x, int y, int z, const mifft::FusedPairArgs* f, unsigned grid, hipStream_t s, int query, int* r0,
                      unsigned* tiles0, unsigned* tiles1);
int mifft_nd2t(int f64, int x, int y, int z, const mifft::TileArgs* a, const mifft::TiledGeom* g, hipStream_t s, int query);
int mifft_nd2t_split_in(int f64, int x, int y, int z, const mifft::TileArgs* a, const mifft::TiledGeom* g, hipStream_t s, int query);
int mifft_nd2t_split(int f64, int x, int y, int z, const mifft::TileArgs* a, const mifft::TiledGeom* g, hipStream_t s, int query);
int mifft_nd2p(int f64, int x, int y, int z, const mifft::TileArgs* a, hipStream_t s, int query);     // fft_nd2p.hip: dense planes, 16-byte accesses
int mifft_nd2zp(int f64, int x, int y, int z, const mifft::TileArgs* a, hipStream_t s, int query);    // fft_nd2zp.hip: the same with several work-groups per transform (out of place)
int mifft_mixed_supported_impl(int f64, int n);
int mifft_mixed_launch(int f64, int n, long long rows, long long stride_in, long long stride_out, long long inner, const void* in,
                       void* out, const void* tw, int flags, double scale, hipStream_t s);
int mifft_mixed_long_split_impl(int f64, long long n, int* n1, int* n2);
int mifft_mixed_long_launch(int f64, int n1, int n2, long long batch, const void* in, void* mid, void* out, const void* tw1, const void* tw2,
                            const void* tw_lo, const void* tw_hi, int tw_shift, int flags, double scale, hipStream_t s);
int mifft_bluestein_padded_impl(int f64, int n);
int mifft_bluestein_len_supported_impl(int f64, int m);
int mifft_mixed_nd_supported_impl(int f64, int nx, int ny, int nz);
int mifft_mixed_nd_rows_ok_impl(int f64, int n);
int mifft_mixed_nd_launch(int f64, int nx, int ny, int nz, long long transforms, const void* in, void* out, const void* twx, const void* twy,
                          const void* twz, int flags, double scale, hipStream_t s);
int mifft_bluestein_launch(int f64, int n, int m, long long rows, long long stride_in, long long stride_out, const void* in, void* out,
                           const void* tw, const void* chirp, const void* bhat, int flags, double scale, hipStream_t s);
int mifft_xcd2_f32_launch(const mifft::Xcd2Args* f, int split, int prefetch, int mode, unsigned grid, hipStream_t s);
// fft_nd2z.hip: 0 = launched (query 1: a kernel exists; query 2: one that is preferred at every buffer size), -2 = none, -1 = grid too large
int mifft_nd2z(int f64, int x, int y, int z, const mifft::TileArgs* a, hipStream_t s, int query);
}

namespace mifft {
template <typename T, int L, int W, int NT, bool ROW, bool TR, typename RL>
static inline int launch_tile(const TileArgs* a, hipStream_t s, int query_only) {
    if (query_only) return 0;
    const long long tiles = (a->total + W - 1) / W;
    if (tiles <= 0) return 0;
    if (tiles > 2147483647ll) return -1;
    hipLaunchKernelGGL((fft_tile_kernel<T, L, W, NT, ROW, TR, RL>), dim3((unsigned)tiles), dim3(NT), 0, s, *a);
    return (int)hipGetLastError();
}
}  // namespace mifft

#define MIFFT_COL_CASE(T, Lv, Wv, NTv, ...)                                                             \
    case Lv:                                                                                            \
        return tr ? mifft::launch_tile<T, Lv, Wv, NTv, false, true, mifft::RadixList<__VA_ARGS__>>(a, s, query_only) \
                  : mifft::launch_tile<T, Lv, Wv, NTv, false, false, mifft::RadixList<__VA_ARGS__>>(a, s, query_only);
#define MIFFT_ROW_CASE(T, Lv, Wv, NTv, ...)                                                             \
    case Lv:                                                                                            \
        return mifft::launch_tile<T, Lv, Wv, NTv, true, false, mifft::RadixList<__VA_ARGS__>>(a, s, query_only);

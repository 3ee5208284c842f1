// Wave-autonomous small transforms for gfx950: N = 4 ... 32 (fp32) / 4 ... 16 (fp64) along the contiguous axis and the
// (16, 16) fp32 plane, with NO LDS and NO barrier.  Counterpart of the reference's "several transforms per work-group"
// small-n local kernel (pyfft/kernel_helpers.py:31-35, kernel.mako:725-803), whose LDS exchange becomes cross-lane
// register moves (DPP: quad_perm / row_shl / row_shr / row_ror) inside one 64-wide wavefront.
//
// One wave owns 64 consecutive transforms (G * 1 KiB of memory, G = lanes per transform).  Loads and stores are whole
// 16-byte pieces, lane-contiguous (1 KiB per wave instruction), so a transform arrives spread over G = N * sizeof(cplx) / 16
// lanes and a lane holds G pieces of G different transforms.  A G x G transposition among the G lanes of a group
// (log2 G steps; step m: lanes l and l ^ m swap the pieces whose index differs in bit m) gives every lane ONE whole
// transform in natural order; the butterflies are the plain in-register Dft<N>; the same transposition leads back to the
// coalesced layout.  No twiddle table is read at all (N <= 32 is one radix).  Per lane and transform: 2 * G * 4 * log2(G)
// * ~2 move/select operations around ~N * log2(N) * 3 flops, an order of magnitude under the VALU roof of a streaming
// kernel; what it buys is occupancy (no LDS, ~40-110 VGPRs) and no barrier, i.e. short ramps on small buffers.
#pragma once
#include "fft_tile.hpp"

namespace mifft {

typedef int wave_i4 __attribute__((ext_vector_type(4)));
// (by value on purpose: __builtin_bit_cast applied directly to a vector-element lvalue reads element 0 whatever the index)
__device__ __forceinline__ float wave_f(int x) { return __builtin_bit_cast(float, x); }
__device__ __forceinline__ int wave_i(float x) { return __builtin_bit_cast(int, x); }

// One exchange step of the transposition for one dword pair (lo = piece j, hi = piece j | MASK): lanes with bit MASK clear
// keep lo and take the partner's lo as their hi; lanes with the bit set keep hi and take the partner's hi as their lo.
// Two instructions per pair: MASK 4 / 8 write only the receiving banks of a DPP move (bank_mask), MASK 1 / 2 are a DPP
// move each that the compiler folds into the select (v_cndmask_b32_dpp).
template <int MASK> __device__ __forceinline__ void wave_swap(int& lo, int& hi, const bool up) {
    static_assert(MASK == 1 || MASK == 2 || MASK == 4 || MASK == 8, "DPP exchange within a row of 16 lanes");
    const int l = lo, h = hi;
    if constexpr (MASK == 4) {
        lo = __builtin_amdgcn_update_dpp(l, h, 0x114, 0xF, 0xA, false);   // row_shr:4, banks 1 and 3: lo <- (lane - 4).hi
        hi = __builtin_amdgcn_update_dpp(h, l, 0x104, 0xF, 0x5, false);   // row_shl:4, banks 0 and 2: hi <- (lane + 4).lo
    } else if constexpr (MASK == 8) {
        lo = __builtin_amdgcn_update_dpp(l, h, 0x128, 0xF, 0xC, false);   // row_ror:8, banks 2 and 3
        hi = __builtin_amdgcn_update_dpp(h, l, 0x128, 0xF, 0x3, false);   // row_ror:8, banks 0 and 1
    } else {
        constexpr int ctrl = MASK == 1 ? 0xB1 : 0x4E;                     // quad_perm:[1,0,3,2] / [2,3,0,1]
        const int ph = __builtin_amdgcn_update_dpp(0, h, ctrl, 0xF, 0xF, false);
        const int pl = __builtin_amdgcn_update_dpp(0, l, ctrl, 0xF, 0xF, false);
        lo = up ? ph : l;
        hi = up ? h : pl;
    }
}

// G x G transposition of 16-byte pieces among the G lanes of each aligned group: a[j] of lane m <-> a[m] of lane j
template <int G, int MASK = 1> __device__ __forceinline__ void wave_transpose(wave_i4* a, const int lane) {
    if constexpr (MASK < G) {
        const bool up = (lane & MASK) != 0;
        static_for<G>([&](auto jj) {
            constexpr int j = jj;
            if constexpr ((j & MASK) == 0) {
                static_for<4>([&](auto dd) {
                    constexpr int d = dd;
                    int lo = a[j][d], hi = a[j | MASK][d];
                    wave_swap<MASK>(lo, hi, up);
                    a[j][d] = lo;
                    a[j | MASK][d] = hi;
                });
            }
        });
        wave_transpose<G, MASK * 2>(a, lane);
    }
}

struct WaveArgs {
    const void* in;
    void* out;
    long long pieces;   // 16-byte pieces in all (= transforms * G)
    int inverse;
    int nt;             // bit 0 / 1: non-temporal loads / stores
    double scale;
};

// ---- 1-D: N points along the contiguous axis, dense rows
template <typename T, int N>
__global__ void __launch_bounds__(256) fft_wave_kernel(const WaveArgs a) {
    constexpr int V = 16 / (int)sizeof(cplx<T>);   // points per piece: 2 (fp32) or 1 (fp64)
    constexpr int G = N / V;                       // lanes per transform = pieces per lane
    static_assert(G >= 2 && G <= 16 && G * V == N, "one DPP row holds a whole group");
    const int lane = threadIdx.x & 63;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    const long long chunks = (a.pieces + 64 * G - 1) / (64 * G);
    const T sx = (T)a.scale;
    const T sy = a.inverse ? -sx : sx;
    const T csign = a.inverse ? (T)-1 : (T)1;
    const wave_i4* src = reinterpret_cast<const wave_i4*>(a.in);
    wave_i4* dst = reinterpret_cast<wave_i4*>(a.out);
    // U chunks per step, so that every lane keeps >= 8 sixteen-byte loads in flight also for the shortest transforms
    constexpr int U = G >= 8 ? 1 : 8 / G;
    for (long long ch = wave * U; ch < chunks; ch += nwaves * U) {
        wave_i4 w[U][G];
        static_for<U>([&](auto uu) {
            constexpr int u = uu;
            static_for<G>([&](auto jj) {
                constexpr int j = jj;
                const long long p = (ch + u) * (64 * G) + lane + j * 64;
                wave_i4 z = {0, 0, 0, 0};
                if (p < a.pieces) z = (a.nt & 1) ? __builtin_nontemporal_load(src + p) : src[p];
                w[u][j] = z;
            });
        });
        static_for<U>([&](auto uu) {
            constexpr int u = uu;
            wave_transpose<G>(w[u], lane);
            cplx<T> v[N];
            static_for<G>([&](auto jj) {
                constexpr int j = jj;
                if constexpr (V == 2) {
                    v[2 * j].x = wave_f(w[u][j][0]);
                    v[2 * j].y = wave_f(w[u][j][1]) * csign;
                    v[2 * j + 1].x = wave_f(w[u][j][2]);
                    v[2 * j + 1].y = wave_f(w[u][j][3]) * csign;
                } else {
                    typedef int i2 __attribute__((ext_vector_type(2)));
                    i2 re = {w[u][j][0], w[u][j][1]}, im = {w[u][j][2], w[u][j][3]};
                    v[j].x = __builtin_bit_cast(double, re);
                    v[j].y = __builtin_bit_cast(double, im) * csign;
                }
            });
            Dft<N, T>::run(v);
            static_for<G>([&](auto jj) {
                constexpr int j = jj;
                if constexpr (V == 2) {
                    w[u][j][0] = wave_i(v[2 * j].x * sx);
                    w[u][j][1] = wave_i(v[2 * j].y * sy);
                    w[u][j][2] = wave_i(v[2 * j + 1].x * sx);
                    w[u][j][3] = wave_i(v[2 * j + 1].y * sy);
                } else {
                    typedef int i2 __attribute__((ext_vector_type(2)));
                    const i2 re = __builtin_bit_cast(i2, v[j].x * sx), im = __builtin_bit_cast(i2, v[j].y * sy);
                    w[u][j][0] = re[0]; w[u][j][1] = re[1]; w[u][j][2] = im[0]; w[u][j][3] = im[1];
                }
            });
            wave_transpose<G>(w[u], lane);
            static_for<G>([&](auto jj) {
                constexpr int j = jj;
                const long long p = (ch + u) * (64 * G) + lane + j * 64;
                if (p < a.pieces) {
                    if (a.nt & 4) store_vec_wt(dst + p, w[u][j]);
                    else if (a.nt & 2) __builtin_nontemporal_store(w[u][j], dst + p);
                    else dst[p] = w[u][j];
                }
            });
        });
    }
}

// ---- 2-D: (16, 16) fp32 planes.  One wave owns 8 planes (16 KiB): lane = (plane t = lane / 8, column pair c = lane % 8),
// load j = row j (eight 128-byte segments per wave instruction).  The y transforms run in the lane (it holds columns
// 2c and 2c+1 complete), then two 8 x 8 transpositions (rows 0-7 and 8-15) give the lane rows c and 8 + c complete for
// the x transforms, and the same transpositions lead back.
template <typename T>
__global__ void __launch_bounds__(256) fft_wave_16x16_kernel(const WaveArgs a) {
    static_assert(sizeof(T) == 4, "fp32 planes");
    const int lane = threadIdx.x & 63;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    const long long planes = a.pieces / 128;
    const long long chunks = (planes + 7) / 8;
    const T sx = (T)a.scale;
    const T sy = a.inverse ? -sx : sx;
    const T csign = a.inverse ? (T)-1 : (T)1;
    const wave_i4* src = reinterpret_cast<const wave_i4*>(a.in);
    wave_i4* dst = reinterpret_cast<wave_i4*>(a.out);
    for (long long ch = wave; ch < chunks; ch += nwaves) {
        const long long plane = ch * 8 + (lane >> 3);
        const bool live = plane < planes;
        const long long p0 = plane * 128 + (lane & 7);
        wave_i4 w[16];
        static_for<16>([&](auto jj) {
            constexpr int j = jj;
            wave_i4 z = {0, 0, 0, 0};
            if (live) z = (a.nt & 1) ? __builtin_nontemporal_load(src + p0 + j * 8) : src[p0 + j * 8];
            w[j] = z;
        });
        // y: two columns of 16 points, in the lane
        static_for<2>([&](auto ee) {
            constexpr int e = ee;
            cplx<T> v[16];
            static_for<16>([&](auto jj) {
                constexpr int j = jj;
                v[j].x = wave_f(w[j][2 * e]);
                v[j].y = wave_f(w[j][2 * e + 1]) * csign;
            });
            Dft<16, T>::run(v);
            static_for<16>([&](auto jj) {
                constexpr int j = jj;
                w[j][2 * e] = wave_i(v[j].x);
                w[j][2 * e + 1] = wave_i(v[j].y);
            });
            __builtin_amdgcn_sched_barrier(0);
        });
        // x: rows c and 8 + c, after the transpositions
        static_for<2>([&](auto hh) {
            constexpr int h = hh;
            wave_transpose<8>(w + 8 * h, lane);
            cplx<T> v[16];
            static_for<8>([&](auto jj) {
                constexpr int j = jj;
                v[2 * j].x = wave_f(w[8 * h + j][0]);
                v[2 * j].y = wave_f(w[8 * h + j][1]);
                v[2 * j + 1].x = wave_f(w[8 * h + j][2]);
                v[2 * j + 1].y = wave_f(w[8 * h + j][3]);
            });
            Dft<16, T>::run(v);
            static_for<8>([&](auto jj) {
                constexpr int j = jj;
                w[8 * h + j][0] = wave_i(v[2 * j].x * sx);
                w[8 * h + j][1] = wave_i(v[2 * j].y * sy);
                w[8 * h + j][2] = wave_i(v[2 * j + 1].x * sx);
                w[8 * h + j][3] = wave_i(v[2 * j + 1].y * sy);
            });
            wave_transpose<8>(w + 8 * h, lane);
            __builtin_amdgcn_sched_barrier(0);
        });
        static_for<16>([&](auto jj) {
            constexpr int j = jj;
            if (live) {
                if (a.nt & 4) store_vec_wt(dst + p0 + j * 8, w[j]);
                else if (a.nt & 2) __builtin_nontemporal_store(w[j], dst + p0 + j * 8);
                else dst[p0 + j * 8] = w[j];
            }
        });
    }
}

}  // namespace mifft

// Whole small 2-D / 3-D transforms inside LDS: one work-group holds P points = P / (x*y*z) complete transforms,
// runs every radix stage of every axis on the LDS copy and writes the result back -- ONE HBM round trip instead of
// one per axis (the reference launches 2-5 kernels for these shapes; its published table has
// (16,16), (16,16,16), (8,8,64) among its 11 shapes, doc/source/index.rst:357-373).
//
// Pass algebra per axis = the Stockham recursion of fft_tile.hpp (decimation in time inside LDS, autosort position
// idxD = (jb & ~(Ns-1))*R + (jb & (Ns-1))), with the axis' elements S = (product of faster axes) apart; the
// transforms of a tile and the slower axes are just more independent "outer" butterflies.  All sizes are runtime
// (shifts), only the radix of a stage is a template parameter, so any power-of-two shape with x*y*z <= P works.
//
// Register edge: when the last stage of all writes runs of >= 128 bytes (the faster axes are contiguous under it), its
// results go straight from registers to HBM (edge_out) instead of through LDS and a linear store: one LDS write + read
// of the whole tile and a barrier less (+2...7 points of roofline).  The mirror image for the first stage measured
// slower than the 16-byte linear load + LDS round it replaces (8-byte accesses in 128-byte runs) and is not built.
#pragma once
#include "fft_tile.hpp"

namespace mifft {

constexpr int kNdMaxStages = 12;

struct NdArgs {
    const void* in0;
    const void* in1;
    void* out0;
    void* out1;
    const void* tw[3];     // per axis: L entries w(L)^k
    long long total;       // total points = batch * x*y*z
    int logL[3];           // log2 of the axis lengths (x, y, z); 0 for an absent axis
    int logS[3];           // log2 of the element stride of the axis: 0, logx, logx+logy
    int nstages;
    unsigned char st_axis[kNdMaxStages];
    unsigned char st_radix[kNdMaxStages];  // 2, 4, 8 or 16
    unsigned char st_logNs[kNdMaxStages];  // log2 of the product of the axis' earlier radices
    int split;      // input layout
    int split_out;  // output layout
    int inverse;
    int edge_out;   // last stage writes its results straight to HBM (interleaved output only)
    int wt;         // write-through stores of the result (MIFFT_FLAG_WRITE_THROUGH: small launches; the store phase, either layout)
    double scale;
};

__device__ __forceinline__ int nd_pad(int i) { return i + (i >> 4); }

// EDGE (last stage of all): results scaled (and conjugated back for the inverse) straight to HBM; `gout` = first byte of
// the tile, `left` = points from the start of the tile to the end of the data (whole transforms are in or out).
template <typename T, int P, int NT, int R, bool EDGE>
__device__ __forceinline__ void nd_stage(cplx<T>* lds, cplx<T>* v, const cplx<T>* tw, int logL, int logS, int logNs, int tid,
                                         char* gout, long long left, T sx, T sy) {
    constexpr int PPT = P / NT;
    constexpr int NB = PPT / R;
    constexpr int logR = R == 2 ? 1 : R == 4 ? 2 : R == 8 ? 3 : 4;
    const int logLR = logL - logR;  // butterflies per transform along this axis
    const int Ns = 1 << logNs;
    int base[NB], jbs[NB];
    static_for<NB>([&](auto bb) {
        constexpr int b = bb;
        const int bid = b * NT + tid;
        const int j = bid & ((1 << logS) - 1);
        const int t = bid >> logS;
        const int jb = t & ((1 << logLR) - 1);
        const int o = t >> logLR;
        base[b] = (o << (logL + logS)) + j;
        jbs[b] = jb;
        static_for<R>([&](auto kk) {
            constexpr int k = kk;
            v[b * R + k] = lds[nd_pad(base[b] + ((jb + (k << logLR)) << logS))];
        });
        if (logNs > 0) {
            const int ai = (jb & (Ns - 1)) << (logL - logNs - logR);
            if (R >= 4 && logL >= 6) {
                // long axis: one look-up + power tree instead of R-1 look-ups all over a table that does not fit the
                // L1 next to the data stream (same reasoning as fft_tile.hpp)
                cplx<T> t[R];
                t[1] = tw[ai];
                static_for<R - 2>([&](auto kk) {
                    constexpr int k = kk + 2;
                    if constexpr ((k & 1) == 0) t[k] = cmul<T>(t[k / 2], t[k / 2]);
                    else t[k] = cmul<T>(t[k - 1], t[1]);
                });
                static_for<R - 1>([&](auto kk) {
                    constexpr int k = kk + 1;
                    v[b * R + k] = cmul<T>(v[b * R + k], t[k]);
                });
            } else {
                static_for<R - 1>([&](auto kk) {
                    constexpr int k = kk + 1;
                    v[b * R + k] = cmul<T>(v[b * R + k], tw[k * ai]);
                });
            }
        }
        Dft<R, T>::run(v + b * R);
    });
    __syncthreads();
    static_for<NB>([&](auto bb) {
        constexpr int b = bb;
        const int jb = jbs[b];
        const int idxD = ((jb & ~(Ns - 1)) << logR) + (jb & (Ns - 1));
        if constexpr (EDGE) {
            if (base[b] < left) {
                static_for<R>([&](auto kk) {
                    constexpr int k = kk;
                    cplx<T> p = v[b * R + k];
                    p.x *= sx;
                    p.y *= sy;
                    const unsigned idx = (unsigned)(base[b] + ((idxD + (k << logNs)) << logS));
                    *reinterpret_cast<cplx<T>*>(gout + idx * (unsigned)sizeof(cplx<T>)) = p;
                });
            }
        } else {
            static_for<R>([&](auto kk) {
                constexpr int k = kk;
                lds[nd_pad(base[b] + ((idxD + (k << logNs)) << logS))] = v[b * R + k];
            });
        }
    });
    if constexpr (!EDGE) __syncthreads();
}

template <typename T, int P, int NT>
__global__ void __launch_bounds__(NT) fft_nd_kernel(const NdArgs a) {
    constexpr int PPT = P / NT;
    static_assert(PPT * NT == P && PPT % 4 == 0, "bad tile");
    __shared__ __attribute__((aligned(16))) cplx<T> lds[P + P / 16];
    const int tid = threadIdx.x;
    const long long g0 = (long long)blockIdx.x * P;  // first point of the tile
    cplx<T> v[PPT];

    // view the I/O helpers of fft_tile.hpp through a TileArgs shell (pointers + layout only)
    TileArgs io;
    io.in0 = a.in0; io.in1 = a.in1; io.out0 = a.out0; io.out1 = a.out1; io.split = a.split; io.split_out = a.split_out;

    const T csign = a.inverse ? (T)-1 : (T)1;
    auto load_phase = [&](auto vv) {
        constexpr int V = vv;
        static_for<PPT / V>([&](auto ii) {
            constexpr int it = ii;
            const int e = (it * NT + tid) * V;
            cplx<T> p[V];
            static_for<V>([&](auto k) { p[k].x = 0; p[k].y = 0; });
            if (g0 + e < a.total) load_vec<T, V>(io, g0 + e, p);
            static_for<V>([&](auto k) { v[V * it + k] = p[k]; });
        });
        static_for<PPT / V>([&](auto ii) {
            constexpr int it = ii;
            const int e = (it * NT + tid) * V;
            static_for<V>([&](auto kk) {
                constexpr int k = kk;
                cplx<T> q = v[V * it + k];
                q.y *= csign;
                lds[nd_pad(e + k)] = q;
            });
        });
    };
    if (a.split) load_phase(IC<4>{}); else load_phase(IC<2>{});
    __syncthreads();

    const T sx = (T)a.scale;
    const T sy = a.inverse ? -sx : sx;
    const long long left = a.total - g0;
    char* gout = reinterpret_cast<char*>(reinterpret_cast<cplx<T>*>(a.out0) + g0);
    for (int s = 0; s < a.nstages; ++s) {
        const int ax = a.st_axis[s];
        const cplx<T>* tw = reinterpret_cast<const cplx<T>*>(a.tw[ax]);
        const int logL = a.logL[ax], logS = a.logS[ax], logNs = a.st_logNs[s];
        const bool edge = s + 1 == a.nstages && a.edge_out;
#define MIFFT_ND_STAGE(R)                                                                          \
    if (edge) nd_stage<T, P, NT, R, true>(lds, v, tw, logL, logS, logNs, tid, gout, left, sx, sy); \
    else nd_stage<T, P, NT, R, false>(lds, v, tw, logL, logS, logNs, tid, gout, left, sx, sy);
        switch (a.st_radix[s]) {
            case 2: MIFFT_ND_STAGE(2) break;
            case 4: MIFFT_ND_STAGE(4) break;
            case 8: MIFFT_ND_STAGE(8) break;
            default:
                if constexpr (PPT >= 16) { MIFFT_ND_STAGE(16) }
                break;
        }
#undef MIFFT_ND_STAGE
    }
    if (a.edge_out) return;

    auto store_phase = [&](auto vv, auto ntc) {
        constexpr int V = vv, NTS = ntc;
        static_for<PPT / V>([&](auto ii) {
            constexpr int it = ii;
            const int e = (it * NT + tid) * V;
            cplx<T> p[V];
            static_for<V>([&](auto kk) {
                constexpr int k = kk;
                p[k] = lds[nd_pad(e + k)];
                p[k].x *= sx;
                p[k].y *= sy;
            });
            if (g0 + e < a.total) store_vec<T, V, NTS>(io, g0 + e, p);
        });
    };
    if (a.wt) {
        if (a.split_out) store_phase(IC<4>{}, IC<2>{}); else store_phase(IC<2>{}, IC<2>{});
    } else {
        if (a.split_out) store_phase(IC<4>{}, IC<0>{}); else store_phase(IC<2>{}, IC<0>{});
    }
}

}  // namespace mifft

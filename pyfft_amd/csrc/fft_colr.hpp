// Short strided-axis (COL) passes entirely in registers: L = 4 ... 32.  A thread owns V adjacent columns
// (16 bytes) of the [L][M*S] matrix: L coalesced loads one row pitch apart (a wave covers 1 KiB of every row), ONE radix-L
// butterfly, the inter-pass twiddle if M > 1, L coalesced stores.  No LDS, no barrier -- the generic tile kernel stages the same
// single-radix pass through LDS twice (global -> LDS -> registers -> LDS -> global).  These are the z (or y) passes of 3-D shapes
// with a short slow axis, e.g. the reference's published (16,16,128) and (32,32,128) (doc/source/index.rst:357-373).
// Pass algebra (SURVEY.md 3.3 / pyfft/kernel.mako:805-1047):  out[l][q][j] = scale * w(L*M)^(l*q) * sum_r in[r][l][j] * w(L)^(r*q).
// Plain form only (S >= V; the transposing S == 1 form of a tiny radix does not occur in the planner's factorisations and takes
// the generic kernel), interleaved on both sides.
#pragma once
#include "fft_tile.hpp"

namespace mifft {

template <typename T, int L, int V, bool TW>
__global__ void __launch_bounds__(256) fft_colr_kernel(const TileArgs a) {
    using VT = T __attribute__((ext_vector_type(2 * V)));
    const long long c = ((long long)blockIdx.x * 256 + threadIdx.x) * V;   // first of this thread's V columns
    if (c >= a.total) return;
    const int logMS = a.logMS, logS = a.logS;
    const long long o = c >> logMS, rem = c & ((1ll << logMS) - 1);
    const cplx<T>* in = reinterpret_cast<const cplx<T>*>(a.in0) + (o * a.ostride_in + rem);
    cplx<T> v[V][L];
    auto loads = [&](auto ntc) __attribute__((always_inline)) {
        static_for<L>([&](auto rr) {
            constexpr int r = rr;
            const VT* p = reinterpret_cast<const VT*>(in + ((long long)r << logMS));
            VT q;
            if constexpr (decltype(ntc)::value != 0) q = __builtin_nontemporal_load(p);
            else q = *p;
            static_for<V>([&](auto jj) {
                constexpr int j = jj;
                v[j][r].x = q[2 * j];
                v[j][r].y = q[2 * j + 1];
            });
        });
    };
    if (a.nt & 1) loads(IC<1>{}); else loads(IC<0>{});
    const T csign = a.inverse ? (T)-1 : (T)1;
    static_for<V>([&](auto jj) {
        constexpr int j = jj;
        static_for<L>([&](auto rr) { v[j][rr].y *= csign; });
        Dft<L, T>::run(v[j]);
    });
    const long long l = rem >> logS, jp = rem & ((1ll << logS) - 1);
    if constexpr (TW) {
        const cplx<T>* twlo = reinterpret_cast<const cplx<T>*>(a.tw_lo);
        const cplx<T>* twhi = reinterpret_cast<const cplx<T>*>(a.tw_hi);
        const unsigned lomask = (1u << a.tw_shift) - 1u;
        static_for<L - 1>([&](auto qq) {
            constexpr int q = qq + 1;
            const unsigned e = (unsigned)l * (unsigned)q;
            const cplx<T> w = cmul<T>(twlo[e & lomask], twhi[e >> a.tw_shift]);
            static_for<V>([&](auto jj) { v[jj][q] = cmul<T>(v[jj][q], w); });   // (the V columns share l: S >= V)
        });
    }
    const T sx = (T)a.scale;
    const T sy = a.inverse ? -sx : sx;
    cplx<T>* out = reinterpret_cast<cplx<T>*>(a.out0) + (o * a.ostride_out + ((l * L) << logS) + jp);
    auto stores = [&](auto ntc) __attribute__((always_inline)) {
        static_for<L>([&](auto qq) {
            constexpr int q = qq;
            VT w;
            static_for<V>([&](auto jj) {
                constexpr int j = jj;
                w[2 * j] = v[j][q].x * sx;
                w[2 * j + 1] = v[j][q].y * sy;
            });
            VT* p = reinterpret_cast<VT*>(out + ((long long)q << logS));
            if constexpr (decltype(ntc)::value == 2) store_vec_wt(p, w);
            else if constexpr (decltype(ntc)::value == 1) __builtin_nontemporal_store(w, p);
            else *p = w;
        });
    };
    if (a.nt & 4) stores(IC<2>{}); else if (a.nt & 2) stores(IC<1>{}); else stores(IC<0>{});
}

template <typename T, int L, int V> static inline int launch_colr(const TileArgs* a, hipStream_t s) {
    const long long threads = a->total / V;
    const long long blocks = (threads + 255) / 256;
    if (blocks <= 0) return 0;
    if (blocks > 2147483647ll) return -1;
    if (a->has_tw) hipLaunchKernelGGL((fft_colr_kernel<T, L, V, true>), dim3((unsigned)blocks), dim3(256), 0, s, *a);
    else hipLaunchKernelGGL((fft_colr_kernel<T, L, V, false>), dim3((unsigned)blocks), dim3(256), 0, s, *a);
    return (int)hipGetLastError();
}

}  // namespace mifft

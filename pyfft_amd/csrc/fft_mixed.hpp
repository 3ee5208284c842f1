// Shared pieces of the mixed-radix kernels (fft_mixed.hip: rows / lines / long transforms / Bluestein; fft_mixed_nd.hip: whole smooth
// N-D transforms in one tile): odd-radix and composite butterflies, the run-time radix list of a smooth length.  Everything lives in
// an anonymous namespace: each translation unit gets its own copy.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "../../include/mifft.h"
#include "fft_butterfly.hpp"

namespace {
using namespace mifft;

constexpr int kMaxStages = 12;

// q = a / d for 0 <= a < 2^22, inv = 1.0f / d
__device__ __forceinline__ int fast_div(int a, float inv) { return (int)(((float)a + 0.5f) * inv); }

// odd radices: forward DFT of R points in registers (e = exp(-2 pi i / R) powers as literals)
template <typename T> __device__ __forceinline__ void dft3(cplx<T>* v) {
    const T c = (T)-0.5, s = (T)0.86602540378443864676;
    const cplx<T> t = v[1] + v[2], d = v[1] - v[2];
    const cplx<T> m = {v[0].x + c * t.x, v[0].y + c * t.y};
    v[0] = v[0] + t;
    v[1] = cplx<T>{m.x + s * d.y, m.y - s * d.x};     // m - i s d
    v[2] = cplx<T>{m.x - s * d.y, m.y + s * d.x};
}
template <typename T> __device__ __forceinline__ void dft5(cplx<T>* v) {
    const T c1 = (T)0.30901699437494742410, c2 = (T)-0.80901699437494742410, s1 = (T)0.95105651629515357212, s2 = (T)0.58778525229247312917;
    const cplx<T> a1 = v[1] + v[4], b1 = v[1] - v[4], a2 = v[2] + v[3], b2 = v[2] - v[3];
    const cplx<T> m1 = {v[0].x + c1 * a1.x + c2 * a2.x, v[0].y + c1 * a1.y + c2 * a2.y};
    const cplx<T> m2 = {v[0].x + c2 * a1.x + c1 * a2.x, v[0].y + c2 * a1.y + c1 * a2.y};
    const cplx<T> n1 = {s1 * b1.x + s2 * b2.x, s1 * b1.y + s2 * b2.y};
    const cplx<T> n2 = {s2 * b1.x - s1 * b2.x, s2 * b1.y - s1 * b2.y};
    v[0] = v[0] + a1 + a2;
    v[1] = cplx<T>{m1.x + n1.y, m1.y - n1.x};          // m1 - i n1
    v[4] = cplx<T>{m1.x - n1.y, m1.y + n1.x};
    v[2] = cplx<T>{m2.x + n2.y, m2.y - n2.x};
    v[3] = cplx<T>{m2.x - n2.y, m2.y + n2.x};
}
template <typename T> __device__ __forceinline__ void dft7(cplx<T>* v) {
    const T c1 = (T)0.62348980185873353053, c2 = (T)-0.22252093395631440429, c3 = (T)-0.90096886790241912624;
    const T s1 = (T)0.78183148246802980871, s2 = (T)0.97492791218182360702, s3 = (T)0.43388373911755812048;
    const cplx<T> a1 = v[1] + v[6], b1 = v[1] - v[6], a2 = v[2] + v[5], b2 = v[2] - v[5], a3 = v[3] + v[4], b3 = v[3] - v[4];
    auto re = [&](T x1, T x2, T x3) { return cplx<T>{v[0].x + x1 * a1.x + x2 * a2.x + x3 * a3.x, v[0].y + x1 * a1.y + x2 * a2.y + x3 * a3.y}; };
    auto im = [&](T y1, T y2, T y3) { return cplx<T>{y1 * b1.x + y2 * b2.x + y3 * b3.x, y1 * b1.y + y2 * b2.y + y3 * b3.y}; };
    const cplx<T> m1 = re(c1, c2, c3), m2 = re(c2, c3, c1), m3 = re(c3, c1, c2);
    const cplx<T> n1 = im(s1, s2, s3), n2 = im(s2, -s3, -s1), n3 = im(s3, -s1, s2);
    v[0] = v[0] + a1 + a2 + a3;
    v[1] = cplx<T>{m1.x + n1.y, m1.y - n1.x};
    v[6] = cplx<T>{m1.x - n1.y, m1.y + n1.x};
    v[2] = cplx<T>{m2.x + n2.y, m2.y - n2.x};
    v[5] = cplx<T>{m2.x - n2.y, m2.y + n2.x};
    v[3] = cplx<T>{m3.x + n3.y, m3.y - n3.x};
    v[4] = cplx<T>{m3.x - n3.y, m3.y + n3.x};
}
// exp(-2 pi i m / n) at compile time (octant reduction in integers, Taylor series on [0, pi / 4]: 1 ulp of double)
struct UnitRoot { double c, s; };
constexpr double series_sin(double x) {
    double t = x, r = x;
    for (int k = 1; k < 14; ++k) { t *= -x * x / ((2.0 * k) * (2.0 * k + 1.0)); r += t; }
    return r;
}
constexpr double series_cos(double x) {
    double t = 1.0, r = 1.0;
    for (int k = 1; k < 14; ++k) { t *= -x * x / ((2.0 * k - 1.0) * (2.0 * k)); r += t; }
    return r;
}
constexpr UnitRoot unit_root(int m, int n) {
    int p = (8 * (m % n)), q = n;                 // angle = 2 pi p / (8 q); 1/8 turn = q
    bool neg_s = false, neg_c = false, swap = false;
    if (p > 4 * q) { p = 8 * q - p; neg_s = true; }
    if (p > 2 * q) { p = 4 * q - p; neg_c = true; }
    if (p > q) { p = 2 * q - p; swap = true; }
    const double x = 6.283185307179586476925286766559 * (double)p / (8.0 * (double)q);
    double c = series_cos(x), s = series_sin(x);
    if (swap) { const double t = c; c = s; s = t; }
    if (neg_c) c = -c;
    if (neg_s) s = -s;
    return UnitRoot{c, -s};
}

template <int R, typename T> __device__ __forceinline__ void dft_any(cplx<T>* v);

// composite radix A * B in registers (natural order in and out): i = i1 + A i2, k = B k1 + k2,
//     X[B k1 + k2] = sum_i1 w(A)^(i1 k1) * [ w(AB)^(i1 k2) * sum_i2 w(B)^(i2 k2) x[i1 + A i2] ]
template <int A, int B, typename T> __device__ __forceinline__ void dft_comp(cplx<T>* v) {
    cplx<T> y[A * B];
    static_for<A>([&](auto ii) {
        constexpr int i1 = ii;
        cplx<T> u[B];
        static_for<B>([&](auto i2) { u[i2] = v[i1 + A * i2]; });
        dft_any<B, T>(u);
        static_for<B>([&](auto kk) {
            constexpr int k2 = kk;
            if constexpr (i1 == 0 || k2 == 0) y[i1 * B + k2] = u[k2];
            else {
                constexpr UnitRoot w = unit_root(i1 * k2, A * B);
                y[i1 * B + k2] = cmul<T>(u[k2], cplx<T>{(T)w.c, (T)w.s});
            }
        });
    });
    static_for<B>([&](auto kk) {
        constexpr int k2 = kk;
        cplx<T> u[A];
        static_for<A>([&](auto i1) { u[i1] = y[i1 * B + k2]; });
        dft_any<A, T>(u);
        static_for<A>([&](auto k1) { v[B * k1 + k2] = u[k1]; });
    });
}

template <int R, typename T> __device__ __forceinline__ void dft_any(cplx<T>* v) {
    if constexpr (R == 3) dft3<T>(v);
    else if constexpr (R == 5) dft5<T>(v);
    else if constexpr (R == 7) dft7<T>(v);
    else if constexpr (R == 6) dft_comp<2, 3, T>(v);
    else if constexpr (R == 9) dft_comp<3, 3, T>(v);
    else if constexpr (R == 10) dft_comp<2, 5, T>(v);
    else if constexpr (R == 12) dft_comp<4, 3, T>(v);
    else if constexpr (R == 14) dft_comp<2, 7, T>(v);
    else if constexpr (R == 15) dft_comp<3, 5, T>(v);
    else Dft<R, T>::run(v);
}

// radix list of n; 0 if n has a prime factor beyond 7.  The fewest stages (every stage is one trip through LDS and one barrier:
// 1000 = 10 * 10 * 10, not 5 * 5 * 5 * 8), then the smallest sum of radices (the least butterfly arithmetic).  Order: by the
// power of two in the radix, odd ones FIRST and 16 last -- the first stage writes with stride R (2 R dwords: 10 / 14 / 30 dwords
// for radix 5 / 7 / 15 spread over the LDS banks, 32 dwords for radix 16 would put a wave on 2 of the 64 banks) and the last
// stage (Ns = n / R) writes consecutive addresses whatever its radix.
constexpr int kRadices[] = {16, 15, 14, 12, 10, 9, 8, 7, 6, 5, 4, 3, 2};

bool search(int n, int depth, int first, int* pick, int at, int sum, int* best, int* best_sum) {
    if (depth == 0) {
        if (n != 1 || sum >= *best_sum) return false;
        *best_sum = sum;
        for (int i = 0; i < at; ++i) best[i] = pick[i];
        return true;
    }
    bool found = false;
    for (int i = first; i < (int)(sizeof(kRadices) / sizeof(int)); ++i) {      // non-increasing radices: combinations, not orders
        const int r = kRadices[i];
        if (n % r) continue;
        pick[at] = r;
        found |= search(n / r, depth - 1, i, pick, at + 1, sum + r, best, best_sum);
    }
    return found;
}

int factor_search(int n, int* radix);

// (the search costs microseconds; a launch loop asks for the same length again and again)
int factor(int n, int* radix) {
    static thread_local int last_n = 0, last_ns = 0, last_radix[kMaxStages];
    if (n != last_n) {
        last_ns = factor_search(n, last_radix);
        last_n = n;
    }
    for (int i = 0; i < last_ns; ++i) radix[i] = last_radix[i];
    return last_ns;
}

int factor_search(int n, int* radix) {
    int m = n;
    for (int c : {2, 3, 5, 7}) while (m % c == 0) m /= c;
    if (n < 2 || m != 1) return 0;
    int pick[kMaxStages], best[kMaxStages];
    for (int depth = 1; depth <= kMaxStages; ++depth) {
        int best_sum = 1 << 30;
        if (!search(n, depth, 0, pick, 0, 0, best, &best_sum)) continue;
        // stable sort by the power of two dividing the radix
        int ns = 0;
        for (int pw = 1; pw <= 16; pw *= 2)
            for (int i = depth - 1; i >= 0; --i)
                if ((best[i] & -best[i]) == pw) radix[ns++] = best[i];
        return ns;
    }
    return 0;
}

#ifndef OCC32
#define OCC32 4
#endif
#ifndef OCC64
#define OCC64 4
#endif
constexpr int kTilePoints32 = 4096, kTilePoints64 = 2048, kThreads = 256;

}  // namespace

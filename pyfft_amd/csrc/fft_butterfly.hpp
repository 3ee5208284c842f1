// Register butterflies for gfx950: forward DFTs of length 2/4/8/16/32 on an array of complex
// registers, natural order in, natural order out.  Everything is indexed at compile time so the
// arrays live in VGPRs (no scratch).
//
// These replace the reference's generated fftKernel2/4/8/16/32 (pyfft/kernel.mako:93-246).  The
// inverse transform is obtained by conjugating on load and on store (see fft_tile.hpp), so only the
// forward sign exists here -- which also sidesteps the reference's direction-blind radix-32
// twiddles (kernel.mako:234-240).
#pragma once
#include <hip/hip_runtime.h>

namespace mifft {

template <typename T> using cplx = T __attribute__((ext_vector_type(2)));

template <int N> struct IC { static constexpr int value = N; constexpr operator int() const { return N; } };

// compile-time unrolled loop: f(IC<0>{}), f(IC<1>{}), ...
template <int I, int N, typename F> __device__ __forceinline__ void static_for_impl(F&& f) {
    if constexpr (I < N) {
        f(IC<I>{});
        static_for_impl<I + 1, N>(f);
    }
}
template <int N, typename F> __device__ __forceinline__ void static_for(F&& f) { static_for_impl<0, N>(f); }

template <typename T> __device__ __forceinline__ cplx<T> cmul(cplx<T> a, cplx<T> b) {
    cplx<T> r;
    r.x = a.x * b.x - a.y * b.y;
    r.y = a.x * b.y + a.y * b.x;
    return r;
}
template <typename T> __device__ __forceinline__ cplx<T> cconj(cplx<T> a) {
    cplx<T> r;
    r.x = a.x;
    r.y = -a.y;
    return r;
}

// cos(pi*k/16), k = 0..8
__device__ constexpr double kCos16[9] = {1.0,
                                         0.98078528040323044912618223613424,
                                         0.92387953251128675612818318939679,
                                         0.83146961230254523707878837761791,
                                         0.70710678118654752440084436210485,
                                         0.55557023301960222474283081394853,
                                         0.38268343236508977172845998403040,
                                         0.19509032201612826784828486847702,
                                         0.0};

constexpr double cos32(int e) {  // cos(2*pi*e/32)
    e &= 31;
    return e <= 8 ? kCos16[e] : e <= 16 ? -kCos16[16 - e] : e <= 24 ? -kCos16[e - 16] : kCos16[32 - e];
}
constexpr double sin32(int e) { return cos32(e - 8); }  // sin(2*pi*e/32)

// v * w(32)^E with w(32) = exp(-2*pi*i/32); trivial rotations cost no multiplies
template <int E, typename T> __device__ __forceinline__ cplx<T> mul_w32(cplx<T> v) {
    constexpr int e = E & 31;
    cplx<T> r;
    if constexpr (e == 0) {
        r = v;
    } else if constexpr (e == 8) {  // -i
        r.x = v.y;
        r.y = -v.x;
    } else if constexpr (e == 16) {
        r.x = -v.x;
        r.y = -v.y;
    } else if constexpr (e == 24) {  // +i
        r.x = -v.y;
        r.y = v.x;
    } else if constexpr (e == 4) {  // (1 - i)/sqrt2
        constexpr T c = (T)kCos16[4];
        r.x = (v.x + v.y) * c;
        r.y = (v.y - v.x) * c;
    } else if constexpr (e == 12) {  // (-1 - i)/sqrt2
        constexpr T c = (T)kCos16[4];
        r.x = (v.y - v.x) * c;
        r.y = -(v.x + v.y) * c;
    } else if constexpr (e == 20) {  // (-1 + i)/sqrt2
        constexpr T c = (T)kCos16[4];
        r.x = -(v.x + v.y) * c;
        r.y = (v.x - v.y) * c;
    } else if constexpr (e == 28) {  // (1 + i)/sqrt2
        constexpr T c = (T)kCos16[4];
        r.x = (v.x - v.y) * c;
        r.y = (v.x + v.y) * c;
    } else {
        constexpr T wr = (T)cos32(e);
        constexpr T wi = (T)(-sin32(e));
        r.x = v.x * wr - v.y * wi;
        r.y = v.x * wi + v.y * wr;
    }
    return r;
}

// cos(2*pi*k/64), k = 0..16
__device__ constexpr double kCos64[17] = {1.0,
                                          0.99518472667219688624483695310948,
                                          0.98078528040323044912618223613424,
                                          0.95694033573220886493579788698027,
                                          0.92387953251128675612818318939679,
                                          0.88192126434835502971275686366039,
                                          0.83146961230254523707878837761791,
                                          0.77301045336273696081090660975847,
                                          0.70710678118654752440084436210485,
                                          0.63439328416364549821517161322549,
                                          0.55557023301960222474283081394853,
                                          0.47139673682599764855638762590525,
                                          0.38268343236508977172845998403040,
                                          0.29028467725446236763619237581740,
                                          0.19509032201612826784828486847702,
                                          0.09801714032956060199419556388864,
                                          0.0};

// v * w(64)^E, w(64) = exp(-2*pi*i/64): E = 16*n + m -> (-i)^n * (cos(t) - i sin(t)), t = 2*pi*m/64
template <int E, typename T> __device__ __forceinline__ cplx<T> mul_w64(cplx<T> v) {
    constexpr int e = E & 63;
    if constexpr ((e & 1) == 0) {
        return mul_w32<e / 2, T>(v);
    } else {
        constexpr int n = e >> 4, m = e & 15;
        constexpr T c = (T)kCos64[m];
        constexpr T sn = (T)kCos64[16 - m];
        cplx<T> r;
        r.x = v.x * c + v.y * sn;
        r.y = v.y * c - v.x * sn;
        return mul_w32<8 * n, T>(r);
    }
}

template <int R, typename T> struct Dft;

template <typename T> struct Dft<1, T> {
    static __device__ __forceinline__ void run(cplx<T>*) {}
};

template <typename T> struct Dft<2, T> {
    static __device__ __forceinline__ void run(cplx<T>* v) {
        cplx<T> a = v[0], b = v[1];
        v[0] = a + b;
        v[1] = a - b;
    }
};

template <typename T> struct Dft<4, T> {
    static __device__ __forceinline__ void run(cplx<T>* v) {
        cplx<T> s0 = v[0] + v[2], d0 = v[0] - v[2];
        cplx<T> s1 = v[1] + v[3], d1 = v[1] - v[3];
        cplx<T> md1 = mul_w32<8, T>(d1);  // -i * d1
        v[0] = s0 + s1;
        v[1] = d0 + md1;
        v[2] = s0 - s1;
        v[3] = d0 - md1;
    }
};

// R = A * B:  X[k1 + A*k2] = sum_b w(R)^(b*k1) [ sum_a v[a*B + b] w(A)^(a*k1) ] w(B)^(b*k2)
template <int R, int A, typename T> struct DftComposite {
    static constexpr int B = R / A;
    static __device__ __forceinline__ void run(cplx<T>* v) {
        cplx<T> u[R];  // u[b*A + k1]
        static_for<B>([&](auto bb) {
            constexpr int b = bb;
            cplx<T> t[A];
            static_for<A>([&](auto aa) {
                constexpr int a = aa;
                t[a] = v[a * B + b];
            });
            Dft<A, T>::run(t);
            static_for<A>([&](auto kk) {
                constexpr int k1 = kk;
                u[b * A + k1] = mul_w32<(b * k1) * (32 / R), T>(t[k1]);
            });
        });
        static_for<A>([&](auto kk) {
            constexpr int k1 = kk;
            cplx<T> s[B];
            static_for<B>([&](auto bb) {
                constexpr int b = bb;
                s[b] = u[b * A + k1];
            });
            Dft<B, T>::run(s);
            static_for<B>([&](auto k2) {
                constexpr int kk2 = k2;
                v[k1 + A * kk2] = s[kk2];
            });
        });
    }
};

template <typename T> struct Dft<8, T> {
    static __device__ __forceinline__ void run(cplx<T>* v) { DftComposite<8, 2, T>::run(v); }
};
template <typename T> struct Dft<16, T> {
    static __device__ __forceinline__ void run(cplx<T>* v) { DftComposite<16, 4, T>::run(v); }
};
template <typename T> struct Dft<32, T> {
    static __device__ __forceinline__ void run(cplx<T>* v) { DftComposite<32, 4, T>::run(v); }
};

}  // namespace mifft

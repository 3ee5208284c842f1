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

// Write-through ("sc1") store of one complex number at (wave-uniform base in SGPRs) + (32-bit per-lane byte offset): the line is
// written to the fabric at once and not kept in the XCD's L2.  For STREAMED output (nobody on this XCD re-reads it) a copy kernel
// measured 6.1 TB/s with these stores against 5.6 TB/s with non-temporal and 5.3 TB/s with plain ones
// (profiles/r03_a_l2_resident_probe.log).  HIP has no builtin for a 16-byte agent-scope store, so the encoding is spelled out; the
// s_nop 1 = the TWO wait states gfx940+ needs between a store of more than 64 bits and a VALU write of its data registers (the
// compiler's hazard recogniser does not look inside inline assembly; with one wait state the nd2 fp64 inverse kernels stored a
// later value in a few lanes).
// 16 bytes, write-through, at (wave-uniform base) + (32-bit per-lane byte offset): a RAW BUFFER store with the sc1 cache-policy bit
// (round 6).  Rounds 2-5 spelled `global_store_dwordx4 ... sc1` out in inline assembly (HIP has no 16-byte agent-scope store): a store
// the compiler does not see is a store its s_waitcnt bookkeeping does not count, so every wait it placed for a LATER load (the table
// look-ups of the next exchange round) was short by the stores in flight and, with one in-order counter per wave, became a wait for
// those stores to be acknowledged.  The buffer instruction is an ordinary machine instruction: counted, scheduled, and covered by the
// hazard recogniser (the hand-placed s_nop of the assembly form is gone).  num_records = 2^32 - 1 bytes with stride 0: no lane of a tile
// is ever out of range (tile-local offsets are 32-bit by construction).
__device__ __forceinline__ void store_b128_sc1(const void* base, unsigned voff, unsigned __attribute__((ext_vector_type(4))) v) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0xffffffffu, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, (int)voff, 0, 16 /* sc1 */);
}

template <typename T> __device__ __forceinline__ void store_wt(char* sbase, unsigned voff, cplx<T> r) {
    if constexpr (sizeof(cplx<T>) == 16) {
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        store_b128_sc1(sbase, voff, __builtin_bit_cast(u4, r));
    } else {
        // (round 6: the 8-byte form is what a relaxed agent-scope atomic store compiles to -- global_store_dwordx2 ... sc1 with the same
        // SGPR base + VGPR offset addressing -- and, unlike an asm statement, the compiler COUNTS it: behind spelled-out stores its
        // s_waitcnt vmcnt(n) for a later load under-counted what is in flight and made the wave wait for the stores as well)
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(sbase + voff), __builtin_bit_cast(unsigned long long, r), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
}

// the same with a per-lane 64-bit address (bases that differ across the wave)
template <typename T> __device__ __forceinline__ void store_wt_ptr(void* p, cplx<T> r) {
    if constexpr (sizeof(cplx<T>) == 16) {
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        const u4 v = __builtin_bit_cast(u4, r);
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    } else {
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), __builtin_bit_cast(unsigned long long, r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ... and of any 8 / 16 / 32-byte vector at a per-lane address (the small-launch store policy, MIFFT_FLAG_WRITE_THROUGH)
template <typename VT> __device__ __forceinline__ void store_vec_wt(VT* p, const VT& w) {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    if constexpr (sizeof(VT) == 8) {
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), __builtin_bit_cast(unsigned long long, w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if constexpr (sizeof(VT) == 16) {
        const u4 v = __builtin_bit_cast(u4, w);
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    } else {
        static_assert(sizeof(VT) == 32, "8, 16 or 32 bytes");
        struct Halves { u4 lo, hi; };
        const Halves h = __builtin_bit_cast(Halves, w);
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(h.lo) : "memory");
        asm volatile("global_store_dwordx4 %0, %1, off offset:16 sc1\n\ts_nop 1" ::"v"(p), "v"(h.hi) : "memory");
    }
}

template <typename T> __device__ __forceinline__ cplx<T> cmul(cplx<T> a, cplx<T> b) {
    cplx<T> r;
    r.x = a.x * b.x - a.y * b.y;
    r.y = a.x * b.y + a.y * b.x;
    return r;
}
// fp32: the packed-math form, two instructions instead of four (v_pk_mul_f32 / v_pk_fma_f32 with operand-half selection and a
// negated low half).  The compiler does not find it: from the vector expression it folds the broadcasts of a.x / a.y but builds
// (-b.y, b.x) with a v_xor and a v_mov, and its SLP vectoriser packs the four products into two half-used instructions.
//     t = (a.x * b.x, a.x * b.y);   r = (-a.y * b.y + t.x, a.y * b.x + t.y)
template <> __device__ __forceinline__ cplx<float> cmul<float>(cplx<float> a, cplx<float> b) {
    cplx<float> t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
    return r;
}
// v * w for a COMPILE-TIME w: the constant sits in an SGPR pair (packed instructions take no literal), so a constant rotation
// is two packed instructions and two scalar moves instead of four vector ones
__device__ __forceinline__ cplx<float> cmul_const(cplx<float> v, cplx<float> w) {
    cplx<float> t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(v), "s"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(v), "s"(w), "v"(t));
    return r;
}
// a + (-i) b and a + i b in one instruction each (the rotation is an operand-half swap plus one negated half)
__device__ __forceinline__ cplx<float> cadd_mi(cplx<float> a, cplx<float> b) {
    cplx<float> r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ cplx<float> cadd_pi(cplx<float> a, cplx<float> b) {
    cplx<float> r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// (-i) v = (v.y, -v.x) and i v = (-v.y, v.x): one packed multiply by (1, -1) / (-1, 1) with the halves of v swapped
__device__ __forceinline__ cplx<float> crot_mi(cplx<float> v) {
    cplx<float> r;
    const cplx<float> c = {1.0f, -1.0f};
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(v), "s"(c));
    return r;
}
__device__ __forceinline__ cplx<float> crot_pi(cplx<float> v) {
    cplx<float> r;
    const cplx<float> c = {-1.0f, 1.0f};
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(v), "s"(c));
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
    if constexpr (sizeof(T) == 4 && (e & 7) != 0) {   // fp32, a real multiplication: packed form
        const cplx<float> w = {(float)cos32(e), (float)(-sin32(e))};
        return cmul_const(v, w);
    } else if constexpr (sizeof(T) == 4 && e == 8) {
        return crot_mi(v);
    } else if constexpr (sizeof(T) == 4 && e == 24) {
        return crot_pi(v);
    } else if constexpr (e == 0) {
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
    } else if constexpr (sizeof(T) == 4) {
        constexpr int n = e >> 4, m = e & 15;   // w(64)^e = (-i)^n (cos t - i sin t), t = 2 pi m / 64
        constexpr double c = kCos64[m], sn = kCos64[16 - m];
        constexpr double wr = n == 0 ? c : n == 1 ? -sn : n == 2 ? -c : sn;
        constexpr double wi = n == 0 ? -sn : n == 1 ? -c : n == 2 ? sn : c;
        const cplx<float> w = {(float)wr, (float)wi};
        return cmul_const(v, w);
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
        v[0] = s0 + s1;
        v[2] = s0 - s1;
        if constexpr (sizeof(T) == 4) {
            v[1] = cadd_mi(d0, d1);   // d0 - i d1
            v[3] = cadd_pi(d0, d1);   // d0 + i d1
        } else {
            cplx<T> md1 = mul_w32<8, T>(d1);  // -i * d1
            v[1] = d0 + md1;
            v[3] = d0 - md1;
        }
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

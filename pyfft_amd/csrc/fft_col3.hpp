// Strided-axis (COL) kernels for L = 2048 in fp32 and L = 1024 in fp64: the two-phase col2 data flow (fft_col2.hpp) at 512
// threads, so that fp32 N = 2^21 / 2^22 and fp64 N = 2^19 / 2^20 need two HBM round trips instead of three (counterpart of
// the reference's four-pass chain for 2^22, pyfft/kernel_helpers.py:67-122).
//
// L = 2 * L' (L' = 256 * A: 1024 in fp32, A = 4; 512 in fp64, A = 2) by decimation in TIME: the even rows and the odd rows
// of a 16-column tile are two independent L'-point col2 transforms (16 * A points per thread, radix 16 * A * 16, one LDS
// exchange each), and
//     X[q'] = E[q'] + w(L)^q' * O[q'],    X[q' + L'] = E[q'] - w(L)^q' * O[q'].
// Lanes 0-31 of every wave work on the even rows, lanes 32-63 on the odd rows, with the same col2 thread index in both
// halves: the two operands of one output pair then sit in lanes l and l ^ 32 of one wave and the combination is one
// v_permlane32_swap per 32-bit register (gfx950) -- no LDS, no barrier.  The halves have their own LDS exchange buffers
// (2 x 34 KiB; 16-byte points go through them one component at a time, as in col2).  One work-group (8 waves, ~190 VGPRs)
// per CU.
//
// Pass algebra as in fft_col2.hpp (SURVEY.md 3.3 / pyfft/kernel.mako:805-1047):
//     out[l][q][j] = scale * w(L*M)^(l*q) * sum_r in[r][l][j] * w(L)^(r*q)
// with r = 2*r' + h, r' = b1*16A + a*16 + b0 and q = h'*L' + qb0*16A + qa*16 + qb1.
#pragma once
#include "fft_col2.hpp"

namespace mifft {

// The radix-2 across the lane halves.  In: every lane's own value (lanes 0-31: E[q'], lanes 32-63: w * O[q'] of the thread 32
// lanes below).  Out: lanes 0-31 X[q'] = E + w O, lanes 32-63 X[q' + L'] = E - w O.  v_permlane32_swap exchanges the upper half
// of its first operand with the lower half of its second, so swapping the REAL register with the IMAGINARY one leaves lane l
// with (E.re, wO.re) and lane l + 32 with (E.im, wO.im): every lane forms the sum and the difference of ONE component, and a
// second swap of (sum, difference) puts (X.re, X.im) into lane l and (X'.re, X'.im) into lane l + 32.  Four instructions per
// point, no copies and no selects.
__device__ __forceinline__ void col3_swap(float& p, float& q) {
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(int, p), __builtin_bit_cast(int, q), false, false);
    p = __builtin_bit_cast(float, (int)r[0]);
    q = __builtin_bit_cast(float, (int)r[1]);
}
__device__ __forceinline__ void col3_swap(double& p, double& q) {
    typedef int i2 __attribute__((ext_vector_type(2)));
    i2 a = __builtin_bit_cast(i2, p), b = __builtin_bit_cast(i2, q);
    const auto r0 = __builtin_amdgcn_permlane32_swap(a[0], b[0], false, false);
    const auto r1 = __builtin_amdgcn_permlane32_swap(a[1], b[1], false, false);
    a[0] = r0[0]; b[0] = r0[1];
    a[1] = r1[0]; b[1] = r1[1];
    p = __builtin_bit_cast(double, a);
    q = __builtin_bit_cast(double, b);
}
template <typename T> __device__ __forceinline__ cplx<T> col3_combine(cplx<T> own) {
    T p = own.x, q = own.y;
    col3_swap(p, q);
    T sum = p + q, dif = p - q;
    col3_swap(sum, dif);
    cplx<T> r;
    r.x = sum;
    r.y = dif;
    return r;
}

// LDS of one work-group in units of T: two halves of BUF slots; 8-byte points store a whole point per slot (2 T),
// 16-byte points one component at a time (1 T per slot)
template <typename T, bool TR> struct Col3Lds {
    static constexpr int PITCH = TR ? 17 : 16;
    static constexpr int BUF = 16 * 16 * PITCH;   // slots per half
    static constexpr bool HALF = sizeof(T) > 4;
    // 8-byte points double-buffer the exchange by slab parity (the work-group owns its CU anyway: 2 x 68 KiB of the 160), which
    // drops the "previous round's reads are done" barrier; 16-byte points already take two barriers per component
    static constexpr bool DOUBLE = !HALF;
    static constexpr int SCALARS = 2 * BUF * (HALF ? 1 : 2) * (DOUBLE ? 2 : 1);
};

// 16-byte agent-scope write-through store of the fp64 tiles: fft_butterfly.hpp store_b128_sc1 (a raw buffer store with the sc1 bit since
// round 6; rounds 2-5 spelled a global store out in assembly, with the wait state gfx9 needs between a store of more than 64 bits and a
// VALU write of its data registers by hand -- without it the first row of every pass-0 tile came out with a later value's real part in 4
// lanes of 16).
__device__ __forceinline__ void col3_store_wt16(const char* base, unsigned voff, const cplx<double>& r) {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    store_b128_sc1(base, voff, __builtin_bit_cast(u4, r));
}

// phase 1 of a tile on its own: v[a*16 + b1] = in[2*(b1*16A + a*16 + b0) + h][column c] -- 16 A loads per thread, all issued before
// anything waits for them.  The persistent kernels of round 6 call it at the END of the previous tile (fft_fused2.hpp,
// fused_list_prefetch), everything else from col3_tile below.
template <typename T, int A, bool SPLIT, bool NTIN>
__device__ __forceinline__ void col3_load(const TileArgs& a, const long long o_in, const long long rem0, cplx<T>* v) {
    constexpr int PPT = 16 * A;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int h = lane >= 32 ? 1 : 0;
    const int tl = (lane & 31) | ((tid >> 6) << 5);
    const int c = tl & 15, b0 = tl >> 4;
    int logMS = a.logMS;
    asm volatile("" : "+s"(logMS));
    const long long ubase = o_in * a.ostride_in + rem0;
    const unsigned voff = ((unsigned)(2 * b0 + h) << logMS) + (unsigned)c;
    if constexpr (!SPLIT) {
        const char* src = reinterpret_cast<const char*>(reinterpret_cast<const cplx<T>*>(a.in0) + ubase);
        const unsigned vb = voff * (unsigned)sizeof(cplx<T>);
        static_for<PPT>([&](auto kk) {
            constexpr int k = kk, ia = k >> 4, b1 = k & 15;
            const char* p = src + (((long long)(2 * (b1 * 16 * A + ia * 16)) << logMS) * (long long)sizeof(cplx<T>));
            if constexpr (NTIN) v[k] = __builtin_nontemporal_load(reinterpret_cast<const cplx<T>*>(p + vb));
            else v[k] = *reinterpret_cast<const cplx<T>*>(p + vb);
        });
    } else {
        const char* sre = reinterpret_cast<const char*>(reinterpret_cast<const T*>(a.in0) + ubase);
        const char* sim = reinterpret_cast<const char*>(reinterpret_cast<const T*>(a.in1) + ubase);
        const unsigned vb = voff * (unsigned)sizeof(T);
        static_for<PPT>([&](auto kk) {
            constexpr int k = kk, ia = k >> 4, b1 = k & 15;
            const long long off = ((long long)(2 * (b1 * 16 * A + ia * 16)) << logMS) * (long long)sizeof(T);
            if constexpr (NTIN) {
                v[k].x = __builtin_nontemporal_load(reinterpret_cast<const T*>(sre + off + vb));
                v[k].y = __builtin_nontemporal_load(reinterpret_cast<const T*>(sim + off + vb));
            } else {
                v[k].x = *reinterpret_cast<const T*>(sre + off + vb);
                v[k].y = *reinterpret_cast<const T*>(sim + off + vb);
            }
        });
    }
}

// The table factors of the two register stages of a tile (they depend on the thread, not on the tile): ONE batch of look-ups, issued by
// the caller right behind the tile's own loads -- a wave's loads return in order, so they arrive with the last rows of the tile.
template <typename T, int A> struct Col3StageTw {
    ColStageTw<T> tw;
    cplx<T> twA[A > 1 ? A - 1 : 1];
    __device__ __forceinline__ void load(const TileArgs& a) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int b0 = (((tid & 63) & 31) | ((tid >> 6) << 5)) >> 4;
        const cplx<T>* twL = reinterpret_cast<const cplx<T>*>(a.tw_L);
        tw.load(twL, b0, 2);          // s = w(L')^b0 = w(L)^(2*b0)
        if constexpr (A > 1) {
            static_for<A - 1>([&](auto qq) {
                constexpr int qa = qq + 1;
                twA[qq] = twL[32 * b0 * qa];          // w(16A)^(b0*qa) = w(L)^(32*b0*qa)
            });
        }
    }
};

// The tile from its loaded operands on: v = the 16 A points per thread col3_load delivers (the caller's registers), st = the stage
// factors Col3StageTw::load delivers.  `hook` runs in
// the middle of the tile (between the register stages and the exchange rounds), `hook2` in the LAST exchange round in front of its
// barrier -- what thread 0 writes to LDS there, every thread may read once the tile is over.
template <typename T, int A, bool TR, bool TW, bool NTOUT, bool SPLIT_OUT, bool WT, typename Hook, typename Hook2>
__device__ __forceinline__ void col3_body(const TileArgs& a, const long long o_out, const long long rem0, T* lds, cplx<T>* v, Col3StageTw<T, A>& st,
                                          Hook hook, Hook2 hook2) {
    constexpr int L = 512 * A;          // 2 * L'
    constexpr int PPT = 16 * A;
    constexpr int PITCH = Col3Lds<T, TR>::PITCH;
    constexpr int BUF = Col3Lds<T, TR>::BUF;
    constexpr bool kHalf = Col3Lds<T, TR>::HALF;

    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const bool upper = lane >= 32;                       // h = 1: the odd rows
    const int h = upper ? 1 : 0;
    const int tl = (lane & 31) | ((tid >> 6) << 5);      // col2 thread index 0..255, the same in both halves
    const int c = tl & 15, b0 = tl >> 4;
    int logMS = a.logMS, logS = a.logS;
    asm volatile("" : "+s"(logMS), "+s"(logS));
    const cplx<T>* twL = reinterpret_cast<const cplx<T>*>(a.tw_L);   // w(L)^k
    const T csign = a.inverse ? (T)-1 : (T)1;
    cplx<T>* const ldsh = reinterpret_cast<cplx<T>*>(lds) + h * BUF;   // whole points (8-byte points)
    T* const ldss = lds + h * BUF;                                      // one component at a time (16-byte points)

    // (Round 6: left to the scheduler the stage look-ups were issued in three groups with a full wait after each, and the look-ups of
    // every exchange round below sat behind the previous round's stores -- a wave then waited for those stores to be acknowledged and
    // for up to six table look-ups ONE AFTER THE OTHER, four times per tile: profiles/r06_fused3_counters.log, waves parked 41 % of
    // their cycles; the ISA sequences are in docs/kernels.md.)
    ColStageTw<T>& tw = st.tw;
    cplx<T>* const twA = st.twA;
    __builtin_amdgcn_sched_barrier(0);
    static_for<PPT>([&](auto kk) { v[kk].y *= csign; });

    // ---- stage 1 (radix-16 over b1) and stage 2 (radix-A over a) of the L'-point half: w(L')^j = w(L)^(2j)
    {
        tw.finish();
        static_for<A>([&](auto aa) {
            constexpr int ia = aa;
            Dft<16, T>::run(v + ia * 16);
            static_for<15>([&](auto q2) {
                constexpr int qb1 = q2 + 1;
                cplx<T> t = v[ia * 16 + qb1];
                if constexpr (ia > 0) t = mul_w16A<A, ia * qb1, T>(t);
                v[ia * 16 + qb1] = cmul<T>(t, tw.template get<qb1>());
            });
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    if constexpr (A > 1) {                     // (A == 1: L = 512 = 2 x 256, no radix-A stage -- the rectangles of round 4)
        static_for<16>([&](auto bb) {
            constexpr int qb1 = bb;
            cplx<T> t[A];
            static_for<A>([&](auto aa) { t[aa] = v[aa * 16 + qb1]; });
            Dft<A, T>::run(t);
            v[qb1] = t[0];
            static_for<A - 1>([&](auto qq) {
                constexpr int qa = qq + 1;
                v[qa * 16 + qb1] = cmul<T>(t[qa], twA[qq]);
            });
        });
    }
    __builtin_amdgcn_sched_barrier(0);
    hook();   // (fft_col2.hpp: the middle of the tile)

    // ---- per slab: exchange + radix-16 of each half, then the radix-2 across the lane halves
    const int u = TR ? (tl & 15) : (tl >> 4);
    const int c2 = TR ? (tl >> 4) : (tl & 15);
    const long long l0 = rem0 >> logS;
    const long long jp0 = rem0 & ((1ll << logS) - 1);
    const unsigned dl = (unsigned)(((rem0 + c2) >> logS) - l0);
    const unsigned djp = (unsigned)(((rem0 + c2) & ((1ll << logS) - 1)) - jp0);
    const unsigned l = (unsigned)l0 + dl;
    const T sx = (T)a.scale;
    const T sy = a.inverse ? -sx : sx;
    cplx<T> sxy;   // one packed multiply per point in fp32
    sxy.x = sx;
    sxy.y = sy;
    const cplx<T>* twlo = reinterpret_cast<const cplx<T>*>(a.tw_lo);
    const cplx<T>* twhi = reinterpret_cast<const cplx<T>*>(a.tw_hi);
    const int tw_shift = a.tw_shift;
    const unsigned lomask = (1u << tw_shift) - 1u;
    // q = h*L' + qb0*16A + qa*16 + u
    const long long oubase = TR ? (a.ostride_out * o_out + rem0 * L) : (a.ostride_out * o_out + ((l0 * L) << logS) + jp0);
    const unsigned ovoff = TR ? ((unsigned)c2 * L + (unsigned)u + (unsigned)(L / 2) * (unsigned)h)
                              : ((((unsigned)dl * L + (unsigned)u + (unsigned)(L / 2) * (unsigned)h) << logS) + djp);

    // The table factors of an exchange round -- w(L)^(qa*16 + u) of the odd half and, with TW, the four anchors of the inter-pass
    // twiddle (two-level table: lo * hi) -- are looked up ONE ROUND AHEAD, in one batch issued before the previous round's stores:
    // the wait for them then covers stores that are a whole round old instead of the sixteen just issued.
    cplx<T> nw0, nlo[TW ? 4 : 1], nhi[TW ? 4 : 1], sstep_lo, sstep_hi;
    auto look_ahead = [&](auto qq) {
        constexpr int qa = qq;
        nw0 = twL[qa * 16 + u];
        if constexpr (TW) {
            static_for<4>([&](auto jj) {
                constexpr int j = jj;
                const unsigned e = l * ((unsigned)(qa * 16 + 64 * A * j) + (unsigned)u + (unsigned)(L / 2) * (unsigned)h);
                nlo[j] = twlo[e & lomask];
                nhi[j] = twhi[e >> tw_shift];
            });
        }
    };
    if constexpr (TW) {
        const unsigned e = l * (16u * A);
        sstep_lo = twlo[e & lomask];
        sstep_hi = twhi[e >> tw_shift];
    }
    look_ahead(IC<0>{});
    __builtin_amdgcn_sched_barrier(0);

    static_for<A>([&](auto rr) {
        constexpr int qa = rr;
        constexpr bool kDouble = Col3Lds<T, TR>::DOUBLE;
        if constexpr (qa > 0 && !kDouble) __syncthreads();   // the previous round's reads are done
        cplx<T> x[16];
        if constexpr (!kHalf) {
            cplx<T>* const buf = ldsh + (kDouble ? (qa & 1) * 2 * BUF : 0);   // (the two halves of a parity are adjacent)
            static_for<16>([&](auto ss) {
                constexpr int qb1 = ss;
                if constexpr (TR) buf[(b0 * 16 + c) * PITCH + qb1] = v[qa * 16 + qb1];
                else buf[(b0 * 16 + qb1) * 16 + c] = v[qa * 16 + qb1];
            });
            if constexpr (qa == A - 1) hook2();
            __syncthreads();
            static_for<16>([&](auto bb) {
                constexpr int bi = bb;
                if constexpr (TR) x[bi] = buf[(bi * 16 + c2) * PITCH + u];
                else x[bi] = buf[(bi * 16 + u) * 16 + c2];
            });
        } else {
            static_for<2>([&](auto cc) {
                constexpr int comp = cc;
                if constexpr (comp == 1) __syncthreads();  // the real parts have been read
                static_for<16>([&](auto ss) {
                    constexpr int qb1 = ss;
                    const T w = comp == 0 ? v[qa * 16 + qb1].x : v[qa * 16 + qb1].y;
                    if constexpr (TR) ldss[(b0 * 16 + c) * PITCH + qb1] = w;
                    else ldss[(b0 * 16 + qb1) * 16 + c] = w;
                });
                if constexpr (qa == A - 1 && comp == 1) hook2();
                __syncthreads();
                static_for<16>([&](auto bb) {
                    constexpr int bi = bb;
                    T w;
                    if constexpr (TR) w = ldss[(bi * 16 + c2) * PITCH + u];
                    else w = ldss[(bi * 16 + u) * 16 + c2];
                    if constexpr (comp == 0) x[bi].x = w; else x[bi].y = w;
                });
            });
        }
        // this round's factors (in flight since the previous round), then the next round's look-ups before this round's stores
        const cplx<T> w0 = nw0;
        cplx<T> anchor[TW ? 4 : 1];
        cplx<T> sstep;
        if constexpr (TW) {
            static_for<4>([&](auto jj) { anchor[jj] = cmul<T>(nlo[jj], nhi[jj]); });
            sstep = cmul<T>(sstep_lo, sstep_hi);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (qa + 1 < A) {
            look_ahead(IC<qa + 1>{});
            __builtin_amdgcn_sched_barrier(0);
        }
        Dft<16, T>::run(x);
        // x[qb0] = E[q'] (lower lanes) or O[q'] (upper lanes), q' = qb0*16A + qa*16 + u.  The odd half is multiplied by
        // w(L)^q' = w(L)^(qa*16 + u) [one look-up] * w(32)^qb0 [constants: L / 16A = 32]; then the radix-2 across the halves.
        {
            static_for<16>([&](auto qq) {
                constexpr int qb0 = qq;
                const cplx<T> t = mul_w32<qb0, T>(cmul<T>(x[qb0], w0));
                cplx<T> own;
                own.x = upper ? t.x : x[qb0].x;
                own.y = upper ? t.y : x[qb0].y;
                x[qb0] = col3_combine<T>(own);
            });
        }
        if constexpr (TW) {
            static_for<4>([&](auto jj) {
                constexpr int j = jj;
                cplx<T> cur = anchor[j];
                static_for<4>([&](auto ii) {
                    constexpr int qb0 = 4 * j + ii;
                    x[qb0] = cmul<T>(x[qb0], cur);
                    if constexpr (ii < 3) cur = cmul<T>(cur, sstep);
                });
            });
        }
        static_for<16>([&](auto qq) {
            constexpr int qb0 = qq;
            const long long gu = TR ? (oubase + qb0 * 16 * A + 16 * qa) : (oubase + ((long long)(qb0 * 16 * A + 16 * qa) << logS));
            const cplx<T> r = x[qb0] * sxy;
            if constexpr (!SPLIT_OUT) {
                char* p = reinterpret_cast<char*>(reinterpret_cast<cplx<T>*>(a.out0) + gu);
                if constexpr (WT) {
                    if constexpr (sizeof(cplx<T>) == 8)
                        __hip_atomic_store(reinterpret_cast<unsigned long long*>(p + ovoff * (unsigned)sizeof(cplx<T>)),
                                           __builtin_bit_cast(unsigned long long, r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else
                        col3_store_wt16(p, ovoff * (unsigned)sizeof(cplx<T>), r);
                } else if constexpr (NTOUT) {
                    __builtin_nontemporal_store(r, reinterpret_cast<cplx<T>*>(p + ovoff * (unsigned)sizeof(cplx<T>)));
                } else {
                    *reinterpret_cast<cplx<T>*>(p + ovoff * (unsigned)sizeof(cplx<T>)) = r;
                }
            } else {
                char* pr = reinterpret_cast<char*>(reinterpret_cast<T*>(a.out0) + gu);
                char* pi = reinterpret_cast<char*>(reinterpret_cast<T*>(a.out1) + gu);
                if constexpr (NTOUT) {
                    __builtin_nontemporal_store(r.x, reinterpret_cast<T*>(pr + ovoff * (unsigned)sizeof(T)));
                    __builtin_nontemporal_store(r.y, reinterpret_cast<T*>(pi + ovoff * (unsigned)sizeof(T)));
                } else {
                    *reinterpret_cast<T*>(pr + ovoff * (unsigned)sizeof(T)) = r.x;
                    *reinterpret_cast<T*>(pi + ovoff * (unsigned)sizeof(T)) = r.y;
                }
            }
        });
    });
}

// WT: write the result with write-through (agent-coherent) stores: the intermediate of the fused two-pass kernel (interleaved)
template <typename T, int A, bool TR, bool TW, bool SPLIT, bool NTIN, bool NTOUT, bool SPLIT_OUT, bool WT = false, typename Hook = TileNoHook>
__device__ __forceinline__ void col3_tile(const TileArgs& a, const long long o_in, const long long o_out, const long long rem0,
                                          T* lds, Hook hook = Hook()) {
    cplx<T> v[16 * A];
    Col3StageTw<T, A> st;
    col3_load<T, A, SPLIT, NTIN>(a, o_in, rem0, v);
    st.load(a);
    col3_body<T, A, TR, TW, NTOUT, SPLIT_OUT, WT>(a, o_out, rem0, lds, v, st, hook, TileNoHook());
}

template <typename T, int A, bool TR, bool TW, bool SPLIT, bool SPLIT_OUT>
__global__ void __launch_bounds__(512, 2) fft_col3_kernel(const TileArgs a) {
    __shared__ __attribute__((aligned(16))) T lds[Col3Lds<T, TR>::SCALARS];
    const long long col0 = (long long)blockIdx.x * 16;
    const long long o = col0 >> a.logMS;
    const long long rem0 = col0 & ((1ll << a.logMS) - 1);
    if constexpr (TR && !SPLIT) {
        if (a.nt & 4) col3_tile<T, A, TR, TW, SPLIT, false, false, SPLIT_OUT, true>(a, o, o, rem0, lds);
        else if (a.nt & 1) col3_tile<T, A, TR, TW, SPLIT, true, false, SPLIT_OUT>(a, o, o, rem0, lds);
        else col3_tile<T, A, TR, TW, SPLIT, false, false, SPLIT_OUT>(a, o, o, rem0, lds);
    } else if constexpr (!TR && !SPLIT_OUT) {
        if (a.nt & 4) col3_tile<T, A, TR, TW, SPLIT, false, false, SPLIT_OUT, true>(a, o, o, rem0, lds);
        else if (a.nt & 2) col3_tile<T, A, TR, TW, SPLIT, false, true, SPLIT_OUT>(a, o, o, rem0, lds);
        else col3_tile<T, A, TR, TW, SPLIT, false, false, SPLIT_OUT>(a, o, o, rem0, lds);
    } else {
        col3_tile<T, A, TR, TW, SPLIT, false, false, SPLIT_OUT>(a, o, o, rem0, lds);
    }
}

}  // namespace mifft

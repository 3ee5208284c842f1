// "Two-phase" strided-axis tiles for fp32, L = 256 / 512, on 32-column tiles: the data flow of fft_col2.hpp with every thread
// owning TWO adjacent columns, so that a lane moves 16 bytes and a row segment is 256 bytes (round 4).
//
// Why: the persistent two-pass kernels run the fp64 sizes 2^16 ... 2^18 at 0.465-0.474 of the roofline and the fp32 sizes of the
// same BYTES at 0.41-0.43 (profiles/r04_l_anchored_twiddles_fp64_mid.log, r04_long_1d_sizes.log): 16 columns of fp64 are
// 256-byte segments moved by 16-byte lanes, 16 columns of fp32 128-byte segments moved by 8-byte lanes.  L = 1024 cannot take the
// wider tile (128 points per thread), L = 256 / 512 can: 32 / 64 points per thread, the register footprint of the 16-column tile
// of L = 512 / 1024.
//
// Algebra, radix split (16 * A * 16), twiddles and LDS exchange are fft_col2.hpp's, applied to both columns of a thread with the
// same table factors (they depend on the row digits only); one LDS buffer of 16 x 16 x 32 points (64 KiB; transposing form with a
// pad: 68 KiB), two work-groups per CU.  Interleaved fp32 only.
#pragma once
#include "fft_col2.hpp"

namespace mifft {

template <bool TR> struct Col2wLds {
    static constexpr int PITCH = TR ? 17 : 16;          // TR: [b0][c < 32][qb1 + pad]; else [b0][qb1][c < 32]
    static constexpr int ELEMS = 16 * 32 * PITCH;        // complex<float> elements of the one exchange buffer
};

// One tile = 32 adjacent columns starting at column rem0 (a multiple of 32) of matrix o_in; result to matrix o_out.
// WT / NTIN / NTOUT / hook as col2_tile.
template <int A, bool TR, bool TW, bool WT, bool NTIN, bool NTOUT, typename Hook = TileNoHook>
__device__ __forceinline__ void col2w_tile(const TileArgs& a, const long long o_in, const long long o_out, const long long rem0,
                                           cplx<float>* lds, Hook hook = Hook()) {
    using T = float;
    typedef float f4 __attribute__((ext_vector_type(4)));
    constexpr int L = A * 256;
    constexpr int PPT = A * 16;
    constexpr int PITCH = Col2wLds<TR>::PITCH;
    static_assert(A == 1 || A == 2, "32-column tiles exist for L = 256 and 512");

    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int c = tid & 15, b0 = tid >> 4;            // phase 1: thread (b0, c) owns columns 2c, 2c + 1
    int logMS = a.logMS, logS = a.logS;
    asm volatile("" : "+s"(logMS), "+s"(logS));
    const cplx<T>* twL = reinterpret_cast<const cplx<T>*>(a.tw_L);
    const T csign = a.inverse ? (T)-1 : (T)1;

    // ---- phase 1: global -> registers, 16 bytes per lane.  v[col][a*16 + b1] = in[b1*16A + a*16 + b0][column 2c + col]
    cplx<T> v[2][PPT];
    {
        const long long ubase = o_in * a.ostride_in + rem0;
        const char* src = reinterpret_cast<const char*>(reinterpret_cast<const cplx<T>*>(a.in0) + ubase);
        const unsigned vb = (((unsigned)b0 << logMS) + 2u * (unsigned)c) * (unsigned)sizeof(cplx<T>);
        static_for<PPT>([&](auto kk) {
            constexpr int k = kk, ia = k >> 4, b1 = k & 15;
            const char* p = src + (((long long)(b1 * 16 * A + ia * 16) << logMS) * (long long)sizeof(cplx<T>));
            f4 q;
            if constexpr (NTIN) q = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p + vb));
            else q = *reinterpret_cast<const f4*>(p + vb);
            v[0][k].x = q.x; v[0][k].y = q.y * csign;
            v[1][k].x = q.z; v[1][k].y = q.w * csign;
        });
    }

    // ---- stages 1 and 2 (registers), the table factors shared by the two columns
    {
        const cplx<T> s1 = twL[b0], s2 = twL[2 * b0], s4 = twL[4 * b0], s8 = twL[8 * b0];
        const cplx<T> s3 = cmul<T>(s1, s2), s5 = cmul<T>(s4, s1), s6 = cmul<T>(s4, s2), s7 = cmul<T>(s4, s3);
        auto tw = [&](auto kk) -> cplx<T> {
            constexpr int k = kk;
            if constexpr (k == 1) return s1;
            else if constexpr (k == 2) return s2;
            else if constexpr (k == 3) return s3;
            else if constexpr (k == 4) return s4;
            else if constexpr (k == 5) return s5;
            else if constexpr (k == 6) return s6;
            else if constexpr (k == 7) return s7;
            else if constexpr (k == 8) return s8;
            else if constexpr (k == 9) return cmul<T>(s8, s1);
            else if constexpr (k == 10) return cmul<T>(s8, s2);
            else if constexpr (k == 11) return cmul<T>(s8, s3);
            else if constexpr (k == 12) return cmul<T>(s8, s4);
            else if constexpr (k == 13) return cmul<T>(s8, s5);
            else if constexpr (k == 14) return cmul<T>(s8, s6);
            else return cmul<T>(s8, s7);
        };
        static_for<A>([&](auto aa) {
            constexpr int ia = aa;
            Dft<16, T>::run(v[0] + ia * 16);
            Dft<16, T>::run(v[1] + ia * 16);
            static_for<15>([&](auto q2) {
                constexpr int qb1 = q2 + 1;
                const cplx<T> w = tw(IC<qb1>{});
                cplx<T> t0 = v[0][ia * 16 + qb1], t1 = v[1][ia * 16 + qb1];
                if constexpr (ia > 0) {
                    t0 = mul_w16A<A, ia * qb1, T>(t0);
                    t1 = mul_w16A<A, ia * qb1, T>(t1);
                }
                v[0][ia * 16 + qb1] = cmul<T>(t0, w);
                v[1][ia * 16 + qb1] = cmul<T>(t1, w);
            });
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    if constexpr (A > 1) {
        cplx<T> twA[A - 1];
        static_for<A - 1>([&](auto qq) {
            constexpr int qa = qq + 1;
            twA[qq] = twL[16 * b0 * qa];
        });
        static_for<2>([&](auto cc) {
            constexpr int col = cc;
            static_for<16>([&](auto bb) {
                constexpr int qb1 = bb;
                cplx<T> t[A];
                static_for<A>([&](auto aa) { t[aa] = v[col][aa * 16 + qb1]; });
                Dft<A, T>::run(t);
                v[col][qb1] = t[0];
                static_for<A - 1>([&](auto qq) {
                    constexpr int qa = qq + 1;
                    v[col][qa * 16 + qb1] = cmul<T>(t[qa], twA[qq]);
                });
            });
        });
    }
    __builtin_amdgcn_sched_barrier(0);
    hook();

    // ---- exchange + stage 3, one qa slab per round
    // phase-2 roles.  non-TR (u = tid >> 4, c2 = tid & 15): lanes along the columns, the thread keeps its column pair 2 c2, 2 c2 + 1
    //                 (16-byte stores, 256-byte row segments).
    //                 TR (u = tid & 15, c2 = tid >> 4): lanes along q; the thread takes columns c2 and c2 + 16 (8-byte stores, runs of
    //                 16 consecutive q per column).
    const int u = TR ? (tid & 15) : (tid >> 4);
    const int c2 = TR ? (tid >> 4) : (tid & 15);
    const long long l0 = rem0 >> logS;
    const long long jp0 = rem0 & ((1ll << logS) - 1);
    const T sx = (T)a.scale;
    const T sy = a.inverse ? -sx : sx;
    cplx<T> sxy;
    sxy.x = sx;
    sxy.y = sy;
    const cplx<T>* twlo = reinterpret_cast<const cplx<T>*>(a.tw_lo);
    const cplx<T>* twhi = reinterpret_cast<const cplx<T>*>(a.tw_hi);
    const int tw_shift = a.tw_shift;
    const unsigned lomask = (1u << tw_shift) - 1u;
    // column of this thread's result `col` within the tile, and what follows from it
    auto tile_col = [&](int col) { return TR ? (c2 + 16 * col) : (2 * c2 + col); };

    static_for<A>([&](auto rr) {
        constexpr int qa = rr;
        if constexpr (qa > 0) __syncthreads();  // the previous round's reads are done
        static_for<16>([&](auto ss) {
            constexpr int qb1 = ss;
            if constexpr (TR) {
                lds[(b0 * 32 + 2 * c) * PITCH + qb1] = v[0][qa * 16 + qb1];
                lds[(b0 * 32 + 2 * c + 1) * PITCH + qb1] = v[1][qa * 16 + qb1];
            } else {   // the column pair as one 16-byte LDS write
                f4 q;
                q.x = v[0][qa * 16 + qb1].x; q.y = v[0][qa * 16 + qb1].y; q.z = v[1][qa * 16 + qb1].x; q.w = v[1][qa * 16 + qb1].y;
                *reinterpret_cast<f4*>(&lds[(b0 * 16 + qb1) * 32 + 2 * c]) = q;
            }
        });
        __syncthreads();
        cplx<T> res[2][16];
        cplx<T> xin[2][16];
        static_for<16>([&](auto bb) {
            constexpr int bi = bb;
            if constexpr (TR) {
                xin[0][bi] = lds[(bi * 32 + c2) * PITCH + u];
                xin[1][bi] = lds[(bi * 32 + c2 + 16) * PITCH + u];
            } else {
                const f4 q = *reinterpret_cast<const f4*>(&lds[(bi * 16 + u) * 32 + 2 * c2]);
                xin[0][bi].x = q.x; xin[0][bi].y = q.y; xin[1][bi].x = q.z; xin[1][bi].y = q.w;
            }
        });
        static_for<2>([&](auto cc) {
            constexpr int col = cc;
            const int tc = tile_col(col);
            cplx<T>* x = xin[col];
            Dft<16, T>::run(x);
            if constexpr (TW) {
                const unsigned l = (unsigned)((rem0 + tc) >> logS);          // row index of this column in the inter-pass twiddle
                auto look = [&](unsigned e) { return cmul<T>(twlo[e & lomask], twhi[e >> tw_shift]); };
                const cplx<T> sstep = look(l * (16u * A));
                static_for<4>([&](auto jj) {
                    constexpr int j = jj;
                    cplx<T> cur = look(l * (unsigned)(qa * 16 + u + 64 * A * j));
                    static_for<4>([&](auto ii) {
                        constexpr int qb0 = 4 * j + ii;
                        x[qb0] = cmul<T>(x[qb0], cur);
                        if constexpr (ii < 3) cur = cmul<T>(cur, sstep);
                    });
                });
            }
            static_for<16>([&](auto qq) { res[col][qq] = x[qq] * sxy; });
        });
        // ---- stores of the round: q = qb0*16A + qa*16 + u
        if constexpr (TR) {
            // out[o][rem0 + col][q]: per column 8-byte stores, consecutive lanes consecutive q
            static_for<2>([&](auto cc) {
                constexpr int col = cc;
                const long long oubase = a.ostride_out * o_out + (rem0 + 16 * col) * L;
                const unsigned ovoff = (unsigned)c2 * L + (unsigned)u;
                static_for<16>([&](auto qq) {
                    constexpr int qb0 = qq;
                    const long long gu = oubase + qb0 * 16 * A + 16 * qa;
                    char* p = reinterpret_cast<char*>(reinterpret_cast<cplx<T>*>(a.out0) + gu);
                    const cplx<T> r = res[col][qb0];
                    if constexpr (WT)
                        __hip_atomic_store(reinterpret_cast<unsigned long long*>(p + ovoff * 8u), __builtin_bit_cast(unsigned long long, r),
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else if constexpr (NTOUT) __builtin_nontemporal_store(r, reinterpret_cast<cplx<T>*>(p + ovoff * 8u));
                    else *reinterpret_cast<cplx<T>*>(p + ovoff * 8u) = r;
                });
            });
        } else {
            // out[o][l][q][jp]: the column pair is adjacent in memory when S >= 2 (jp even): one 16-byte store per row
            const unsigned dl = (unsigned)(((rem0 + 2 * c2) >> logS) - l0);
            const unsigned djp = (unsigned)(((rem0 + 2 * c2) & ((1ll << logS) - 1)) - jp0);
            const long long oubase = a.ostride_out * o_out + ((l0 * L) << logS) + jp0;
            const unsigned ovoff = (((unsigned)dl * L + (unsigned)u) << logS) + djp;
            static_for<16>([&](auto qq) {
                constexpr int qb0 = qq;
                const long long gu = oubase + ((long long)(qb0 * 16 * A + 16 * qa) << logS);
                char* p = reinterpret_cast<char*>(reinterpret_cast<cplx<T>*>(a.out0) + gu);
                f4 r;
                r.x = res[0][qb0].x; r.y = res[0][qb0].y; r.z = res[1][qb0].x; r.w = res[1][qb0].y;
                if constexpr (WT) store_b128_sc1(p, ovoff * 8u, __builtin_bit_cast(unsigned __attribute__((ext_vector_type(4))), r));
                else if constexpr (NTOUT) __builtin_nontemporal_store(r, reinterpret_cast<f4*>(p + ovoff * 8u));
                else *reinterpret_cast<f4*>(p + ovoff * 8u) = r;
            });
        }
    });
}

// plain launch: one 32-column tile per work-group; nt as fft_col2_kernel (bit 0 non-temporal loads, bit 1 non-temporal stores,
// bit 2 write-through stores)
template <int A, bool TR, bool TW>
__global__ void __launch_bounds__(256, 2) fft_col2w_kernel(const TileArgs a) {
    __shared__ __attribute__((aligned(16))) cplx<float> lds[Col2wLds<TR>::ELEMS];
    const long long col0 = (long long)blockIdx.x * 32;
    const long long o = col0 >> a.logMS;
    const long long rem0 = col0 & ((1ll << a.logMS) - 1);
    if (a.nt & 4) col2w_tile<A, TR, TW, true, false, false>(a, o, o, rem0, lds);
    else if (TR && (a.nt & 1)) col2w_tile<A, TR, TW, false, true, false>(a, o, o, rem0, lds);
    else if (!TR && (a.nt & 2)) col2w_tile<A, TR, TW, false, false, true>(a, o, o, rem0, lds);
    else col2w_tile<A, TR, TW, false, false, false>(a, o, o, rem0, lds);
}

}  // namespace mifft

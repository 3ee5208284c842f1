// "Two-phase" strided-axis (COL) kernels for gfx950: L = 16 * A * 16 with A in {1, 2, 4}
// (L = 256, 512, 1024), W = 16 adjacent columns per work-group, 256 threads, A*16 points per thread.
//
// Why a second COL kernel: the generic tile kernel (fft_tile.hpp) keeps the whole L x 16 tile in LDS, which at
// L = 1024 is 128 KiB -> one work-group per CU -> its LDS/butterfly phases cannot overlap another work-group's
// HBM phase (measured 3.3-4.2 TB/s against a 5.3-5.6 TB/s streaming ceiling for the same access pattern).
// Here the first two radix stages run in registers directly on the global loads, only ONE exchange goes through
// LDS (a 2 x 34 KiB double buffer, one radix-16 slab per round), and the last radix-16 stage stores straight from
// registers.  LDS traffic per point drops from 4 writes + 4 reads to 1 + 1, barriers from 8 to A, and two
// work-groups fit a CU so that one computes while the other streams.
//
// Pass algebra (SURVEY.md 3.3 / pyfft/kernel.mako:805-1047), one tile = 16 columns of the [L][M*S] matrix:
//     out[l][q][j] = scale * w(L*M)^(l*q) * sum_r in[r][l][j] * w(L)^(r*q)
// with r = b1*(16A) + a*16 + b0 and q = qb0*(16A) + qa*16 + qb1 (decimation in frequency, Stockham order):
//     stage 1: radix-16 over b1, twiddle w(L)^((a*16 + b0)*qb1) = w(L)^(b0*qb1) * w(16A)^(a*qb1)   [2nd factor constant]
//     stage 2: radix-A  over a,  twiddle w(16A)^(b0*qa)
//     ---- LDS exchange, one qa slab per round: (b0, c) threads -> (qb1, c) threads ----
//     stage 3: radix-16 over b0
// Phase-2 lanes hold 16 consecutive q (qb1), so both the transposing store (S == 1, contiguous in q) and the
// strided store (128-byte column segments) are issued per round straight from registers.
#pragma once
#include "fft_tile.hpp"

namespace mifft {

template <int A, int E, typename T> __device__ __forceinline__ cplx<T> mul_w16A(cplx<T> v) {
    if constexpr (A == 4)
        return mul_w64<E, T>(v);
    else
        return mul_w32<E, T>(v);  // A == 2: w(32)
}

// The 15 table twiddles s^k of a first stage, s = w(L)^i0: four look-ups (k = 1, 2, 4, 8) plus products of at most three of
// them, the high ones formed on demand (saves ~14 VGPRs against 15 look-ups held across the stage).  `step` = table entries per
// unit of i0 (1 when the table is w(L') of the sub-transform, 2 when it is w(2L')).  Shared by fft_col3.hpp and fft_xcd2.hpp;
// col2_tile below spells the same arithmetic out in place.
template <typename T> struct ColStageTw {
    cplx<T> s1, s2, s3, s4, s5, s6, s7, s8;
    __device__ __forceinline__ void init(const cplx<T>* twL, int i0, int step = 1) {
        load(twL, i0, step);
        finish();
    }
    // the same in two steps: the four look-ups (issue them early, in one batch with the caller's other look-ups), then the products
    __device__ __forceinline__ void load(const cplx<T>* twL, int i0, int step = 1) {
        s1 = twL[step * i0]; s2 = twL[2 * step * i0]; s4 = twL[4 * step * i0]; s8 = twL[8 * step * i0];
    }
    __device__ __forceinline__ void finish() {
        s3 = cmul<T>(s1, s2); s5 = cmul<T>(s4, s1); s6 = cmul<T>(s4, s2); s7 = cmul<T>(s4, s3);
    }
    template <int k> __device__ __forceinline__ cplx<T> get() const {
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
    }
};

// ESZ = bytes per complex element.  L = 1024 in fp32 double-buffers the exchange (2 x 34 KiB: its 250-VGPR kernel fits two
// work-groups per CU either way); L = 512 reuses one 34 KiB buffer with one more barrier per round, so that FOUR
// work-groups share a CU (108 VGPRs; pipelined N = 2^18: 36.5 -> 38 %); 16-byte points reuse one 68 KiB buffer (two per CU).
template <int A, bool TR, int ESZ = 8> struct Col2Lds {
    static constexpr int PITCH = TR ? 17 : 16;   // TR: [b0][c][qb1 + pad]; else [b0][qb1][c]
    static constexpr int BUF = 16 * 16 * PITCH;  // complex elements per exchange buffer
    static constexpr bool DOUBLE = A > 2 && ESZ <= 8;
    // 16-byte points: the exchange moves real parts, then imaginary parts, through BUF scalars (34 KiB instead of 68:
    // four work-groups per CU instead of two for L = 256, at ~120 VGPRs)
    static constexpr bool HALF = ESZ > 8;
    static constexpr int ELEMS = HALF ? (BUF + 1) / 2 : (DOUBLE ? 2 : 1) * BUF;
};

// One tile = 16 adjacent columns starting at column rem0 (a multiple of 16) of matrix o_in; the result goes to
// matrix o_out (same index for a plain launch; a scratch-ring slot in the fused two-pass kernel).
// SPLIT / SPLIT_OUT: layout of the input / output side (they differ only when one side is a plan's internal,
//     always interleaved, temp buffer).
// WT: write the result with write-through (agent-coherent, "sc1") stores -- used by the fused kernel for the
//     intermediate so that publishing it needs no release fence (interleaved fp32 only).
// NTIN / NTOUT: non-temporal hint on the input loads / output stores (streamed-once data in the fused kernel).
// `hook` is called once, by every thread, between the second register stage and the exchange rounds -- the middle of the tile:
// the fused kernel issues the early poll of its next item's dependency there.
struct TileNoHook {
    __device__ __forceinline__ void operator()() const {}
};

// WIDE (second batch of round 4; 512-thread work-groups): TWO adjacent 16-column tiles interleaved at LANE level -- thread index p
//     = (b0, c32) with c32 < 32 the column within the 32-column double tile at rem0 (a multiple of 32), so that one wave instruction
//     touches 32 adjacent columns: whole 128-byte lines of an fp32 plane.  Each half (sub = c32 >> 4) runs the ordinary tile on its
//     own LDS slab (lds + sub * ELEMS); the arithmetic and the exchange layout per half are unchanged.
template <typename T, int A, bool TR, bool TW, bool SPLIT, bool WT = false, bool NTIN = false, bool NTOUT = false,
          bool SPLIT_OUT = SPLIT, bool WIDE = false, typename LdsPtr = cplx<T>*, typename Hook = TileNoHook>
__device__ __forceinline__ void col2_tile(const TileArgs& a, const long long o_in, const long long o_out,
                                          const long long rem0, LdsPtr lds, Hook hook = Hook(), const int tid_in = -1) {
    constexpr int L = A * 256;
    constexpr int PPT = A * 16;
    constexpr int PITCH = Col2Lds<A, TR, sizeof(cplx<T>)>::PITCH;
    constexpr int BUF = Col2Lds<A, TR, sizeof(cplx<T>)>::BUF;
    constexpr bool kDoubleBuf = Col2Lds<A, TR, sizeof(cplx<T>)>::DOUBLE;
    constexpr bool kHalf = Col2Lds<A, TR, sizeof(cplx<T>)>::HALF;

    // tid_in: the thread's index within the tile when a bigger work-group runs several tiles side by side (fft_fused2s_kernel)
    int tid = tid_in >= 0 ? tid_in : (int)threadIdx.x;
    asm volatile("" : "+v"(tid));  // same reason as below, for the per-thread (VGPR) address pieces
    // WIDE: cg = the thread's column within the double tile (global addressing), c = within its half (LDS indices)
    const int cg = WIDE ? (tid & 31) : (tid & 15), b0 = WIDE ? (tid >> 5) : (tid >> 4);
    const int c = cg & 15;
    if constexpr (WIDE) lds = lds + (cg >> 4) * Col2Lds<A, TR, sizeof(cplx<T>)>::ELEMS;
    // The shift amounts are laundered through an empty asm so that, when this body sits inside the persistent
    // loop of the fused kernel, the ~130 wave-uniform row offsets derived from them are recomputed per tile
    // (2 SALU ops each) instead of being hoisted out of the loop and spilled (measured: 453 SGPR spills).
    int logMS = a.logMS, logS = a.logS;
    asm volatile("" : "+s"(logMS), "+s"(logS));
    const cplx<T>* twL = reinterpret_cast<const cplx<T>*>(a.tw_L);
    const T csign = a.inverse ? (T)-1 : (T)1;

    // ---- phase 1: global -> registers.  v[a*16 + b1] = in[b1*16A + a*16 + b0][column c]
    // Addresses are (wave-uniform 64-bit base, SGPRs) + (per-thread 32-bit byte offset, one VGPR) so that the
    // A*16 loads in flight do not each hold a 64-bit address pair (the dispatcher guarantees the offset fits).
    cplx<T> v[PPT];
    {
        const long long ubase = o_in * a.ostride_in + rem0;  // uniform, elements
        const unsigned voff = (((unsigned)b0 << logMS) + (unsigned)cg);
        if constexpr (!SPLIT) {
            const char* src = reinterpret_cast<const char*>(reinterpret_cast<const cplx<T>*>(a.in0) + ubase);
            const unsigned vb = voff * (unsigned)sizeof(cplx<T>);
            static_for<PPT>([&](auto kk) {
                constexpr int k = kk, ia = k >> 4, b1 = k & 15;
                const char* p = src + (((long long)(b1 * 16 * A + ia * 16) << logMS) * (long long)sizeof(cplx<T>));
                if constexpr (NTIN)
                    v[k] = __builtin_nontemporal_load(reinterpret_cast<const cplx<T>*>(p + vb));
                else
                    v[k] = *reinterpret_cast<const cplx<T>*>(p + vb);
            });
        } else {
            const char* sre = reinterpret_cast<const char*>(reinterpret_cast<const T*>(a.in0) + ubase);
            const char* sim = reinterpret_cast<const char*>(reinterpret_cast<const T*>(a.in1) + ubase);
            const unsigned vb = voff * (unsigned)sizeof(T);
            static_for<PPT>([&](auto kk) {
                constexpr int k = kk, ia = k >> 4, b1 = k & 15;
                const long long off = ((long long)(b1 * 16 * A + ia * 16) << logMS) * (long long)sizeof(T);
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
    // The table factors of the two register stages, looked up in ONE batch right behind the tile's own loads (a wave's loads return
    // in order: they arrive with the tile's last rows).  Round 6, see fft_col3.hpp: left to the scheduler such look-ups were issued
    // in groups with a full wait behind each.
    // ---- stage 1: radix-16 over b1 (A butterflies), twiddle w(L)^(b0*qb1) [table] * w(16A)^(a*qb1) [constant]
    // The 15 table twiddles s^k, s = w(L)^b0, are four look-ups (k = 1, 2, 4, 8) plus products of at most three
    // of them, and are held only as long as needed (saves ~14 VGPRs against 15 look-ups held across the stage).
    const cplx<T> s1 = twL[b0], s2 = twL[2 * b0], s4 = twL[4 * b0], s8 = twL[8 * b0];
    cplx<T> twA[A > 1 ? A - 1 : 1];
    if constexpr (A > 1) {
        static_for<A - 1>([&](auto qq) {
            constexpr int qa = qq + 1;
            twA[qq] = twL[16 * b0 * qa];
        });
    }
    __builtin_amdgcn_sched_barrier(0);
    static_for<PPT>([&](auto kk) {
        constexpr int k = kk;
        v[k].y *= csign;
    });

    {
        const cplx<T> s3 = cmul<T>(s1, s2), s5 = cmul<T>(s4, s1), s6 = cmul<T>(s4, s2), s7 = cmul<T>(s4, s3);
        auto tw = [&](auto kk) -> cplx<T> {
            constexpr int k = kk;  // s^k, k = 1..15
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
            Dft<16, T>::run(v + ia * 16);
            static_for<15>([&](auto q2) {
                constexpr int qb1 = q2 + 1;
                cplx<T> t = v[ia * 16 + qb1];
                if constexpr (ia > 0) t = mul_w16A<A, ia * qb1, T>(t);
                v[ia * 16 + qb1] = cmul<T>(t, tw(IC<qb1>{}));
            });
            __builtin_amdgcn_sched_barrier(0);  // one butterfly's temporaries at a time
        });
    }
    // ---- stage 2: radix-A over a (16 butterflies), twiddle w(16A)^(b0*qa) = w(L)^(16*b0*qa)
    if constexpr (A > 1) {
        static_for<16>([&](auto bb) {
            constexpr int qb1 = bb;
            cplx<T> t[A];
            static_for<A>([&](auto aa) {
                constexpr int ia = aa;
                t[ia] = v[ia * 16 + qb1];
            });
            Dft<A, T>::run(t);
            v[qb1] = t[0];
            static_for<A - 1>([&](auto qq) {
                constexpr int qa = qq + 1;
                v[qa * 16 + qb1] = cmul<T>(t[qa], twA[qq]);
            });
        });
    }
    __builtin_amdgcn_sched_barrier(0);
    hook();

    // ---- exchange + stage 3, one qa slab per round
    // phase-2 thread roles: non-TR (u = tid>>4, c2 = tid&15): lanes along the columns (128-byte row segments)
    //                       TR     (u = tid&15, c2 = tid>>4): lanes along q (the write is contiguous in q)
    // (WIDE: the same roles within the thread's half; c2g = its column within the double tile)
    const int u = TR ? c : b0;
    const int c2 = TR ? b0 : c;
    const int c2g = WIDE ? ((cg & 16) + c2) : c2;
    // output addressing, again uniform base + 32-bit per-thread offset.  The tile's 16 columns start at rem0
    // (a multiple of 16): l0/jp0 are uniform, (dl, djp) is the per-thread part (dl > 0 only when S < 16).
    const long long l0 = rem0 >> logS;
    const long long jp0 = rem0 & ((1ll << logS) - 1);
    const unsigned dl = (unsigned)(((rem0 + c2g) >> logS) - l0);
    const unsigned djp = (unsigned)(((rem0 + c2g) & ((1ll << logS) - 1)) - jp0);
    const unsigned l = (unsigned)l0 + dl;  // row index of this thread's column in the inter-pass twiddle
    const T sx = (T)a.scale;
    const T sy = a.inverse ? -sx : sx;
    cplx<T> sxy;
    sxy.x = sx;
    sxy.y = sy;
    const cplx<T>* twlo = reinterpret_cast<const cplx<T>*>(a.tw_lo);
    const cplx<T>* twhi = reinterpret_cast<const cplx<T>*>(a.tw_hi);
    const int tw_shift = a.tw_shift;
    const unsigned lomask = (1u << tw_shift) - 1u;
    // q = qb0*16A + qa*16 + u
    // non-TR: out[o][l][q][jp] -> uniform o*ostride + ((l0*L + qconst) << logS) + jp0 ; thread ((dl*L + u) << logS) + djp
    // TR (S == 1): out[o][l][q] -> uniform o*ostride + rem0*L + qconst ; thread c2*L + u
    const long long oubase = TR ? (a.ostride_out * o_out + rem0 * L) : (a.ostride_out * o_out + ((l0 * L) << logS) + jp0);
    const unsigned ovoff = TR ? ((unsigned)c2g * L + (unsigned)u) : ((((unsigned)dl * L + (unsigned)u) << logS) + djp);

    // The anchors of the inter-pass twiddle (TW) are looked up ONE ROUND AHEAD, in one batch issued before the previous round's
    // stores: the wait for them then covers stores that are a whole round old.  (Round 6: issued inside their own round they sat
    // behind the sixteen write-through stores just issued, and a wave waited for those to be acknowledged -- three times per tile.)
    cplx<T> nlo[TW ? 4 : 1], nhi[TW ? 4 : 1], sstep_lo, sstep_hi;
    auto look_ahead = [&](auto qq) {
        constexpr int qa = qq;
        if constexpr (TW) {
            static_for<4>([&](auto jj) {
                constexpr int j = jj;
                const unsigned e = l * (unsigned)(qa * 16 + u + 64 * A * j);
                nlo[j] = twlo[e & lomask];
                nhi[j] = twhi[e >> tw_shift];
            });
        }
    };
    if constexpr (TW) {
        const unsigned e = l * (16u * A);
        sstep_lo = twlo[e & lomask];
        sstep_hi = twhi[e >> tw_shift];
        look_ahead(IC<0>{});
        __builtin_amdgcn_sched_barrier(0);
    }

    static_for<A>([&](auto rr) {
        constexpr int qa = rr;
        LdsPtr buf = lds + (kDoubleBuf ? (qa & 1) * BUF : 0);
        if constexpr (!kDoubleBuf && qa > 0) __syncthreads();  // the previous round's reads are done
        cplx<T> x[16];
        if constexpr (!kHalf) {
            static_for<16>([&](auto ss) {
                constexpr int qb1 = ss;
                if constexpr (TR)
                    buf[(b0 * 16 + c) * PITCH + qb1] = v[qa * 16 + qb1];
                else
                    buf[(b0 * 16 + qb1) * 16 + c] = v[qa * 16 + qb1];
            });
            __syncthreads();
            static_for<16>([&](auto bb) {
                constexpr int bi = bb;
                if constexpr (TR)
                    x[bi] = buf[(bi * 16 + c2) * PITCH + u];
                else
                    x[bi] = buf[(bi * 16 + u) * 16 + c2];
            });
        } else {
            // one component at a time through the same slots viewed as scalars
            T* sb = reinterpret_cast<T*>(&lds[0]);
            static_for<2>([&](auto cc) {
                constexpr int comp = cc;
                if constexpr (comp == 1) __syncthreads();  // the real parts have been read
                static_for<16>([&](auto ss) {
                    constexpr int qb1 = ss;
                    const T w = comp == 0 ? v[qa * 16 + qb1].x : v[qa * 16 + qb1].y;
                    if constexpr (TR)
                        sb[(b0 * 16 + c) * PITCH + qb1] = w;
                    else
                        sb[(b0 * 16 + qb1) * 16 + c] = w;
                });
                __syncthreads();
                static_for<16>([&](auto bb) {
                    constexpr int bi = bb;
                    T w;
                    if constexpr (TR)
                        w = sb[(bi * 16 + c2) * PITCH + u];
                    else
                        w = sb[(bi * 16 + u) * 16 + c2];
                    if constexpr (comp == 0) x[bi].x = w; else x[bi].y = w;
                });
            });
        }
        cplx<T> anchor[TW ? 4 : 1];
        cplx<T> sstep;
        if constexpr (TW) {
            // this round's anchors (in flight since the previous round), then the next round's look-ups before this round's stores
            static_for<4>([&](auto jj) { anchor[jj] = cmul<T>(nlo[jj], nhi[jj]); });
            sstep = cmul<T>(sstep_lo, sstep_hi);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (qa + 1 < A) {
                look_ahead(IC<qa + 1>{});
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        Dft<16, T>::run(x);
        if constexpr (TW) {
            // w(L*M)^(l*q) for q = qb0*16A + qlow, qlow = qa*16 + u.  Four anchors (qb0 = 0, 4, 8, 12) come from
            // the two-level table (tw_lo * tw_hi); the three factors after each anchor are one multiplication by
            // s = w^(l*16A) each.  Depth <= 3 keeps the fp32 twiddle error at ~2.5e-7 max while only ~10 VGPRs are
            // live (15 table factors held at once made this kernel spill).
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
            // uniform part of the output element index
            const long long gu = TR ? (oubase + qb0 * 16 * A + 16 * qa)
                                    : (oubase + ((long long)(qb0 * 16 * A + 16 * qa) << logS));
            const cplx<T> r = x[qb0] * sxy;   // (one packed multiply in fp32)
            if constexpr (!SPLIT_OUT) {
                char* p = reinterpret_cast<char*>(reinterpret_cast<cplx<T>*>(a.out0) + gu);
                if constexpr (WT) {
                    if constexpr (sizeof(cplx<T>) == 8)
                        __hip_atomic_store(reinterpret_cast<unsigned long long*>(p + ovoff * (unsigned)sizeof(cplx<T>)),
                                           __builtin_bit_cast(unsigned long long, r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else
                        store_wt<T>(p, ovoff * (unsigned)sizeof(cplx<T>), r);
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

// TR: S == 1 (first pass of a long contiguous axis; the store is a transposition)
// TW: multiply by the inter-pass twiddle w(L*M)^(l*q)
template <typename T, int A, bool TR, bool TW, bool SPLIT, bool SPLIT_OUT = SPLIT>
__global__ void __launch_bounds__(256, 2) fft_col2_kernel(const TileArgs a) {
    __shared__ __attribute__((aligned(16))) cplx<T> lds[Col2Lds<A, TR, sizeof(cplx<T>)>::ELEMS];
    const long long col0 = (long long)blockIdx.x * 16;
    const long long o = col0 >> a.logMS;
    const long long rem0 = col0 & ((1ll << a.logMS) - 1);
    // streaming hints (MIFFT_FLAG_STREAM_*): a transposing pass is the first pass of a plan (its input is read once), a
    // plain one the last pass of an axis (when it is the plan's last, nobody re-reads its output)
    // (MIFFT_FLAG_WRITE_THROUGH, small launches: write-through stores whatever the other hints say)
    if constexpr (TR && !SPLIT) {
        if (a.nt & 4) col2_tile<T, A, TR, TW, SPLIT, true, false, false, SPLIT_OUT>(a, o, o, rem0, lds);
        else if (a.nt & 1) col2_tile<T, A, TR, TW, SPLIT, false, !SPLIT, false, SPLIT_OUT>(a, o, o, rem0, lds);   // (planes: plain loads, as measured in rounds 1-3)
        else col2_tile<T, A, TR, TW, SPLIT, false, false, false, SPLIT_OUT>(a, o, o, rem0, lds);
    } else if constexpr (!TR && !SPLIT_OUT) {
        if (a.nt & 4) col2_tile<T, A, TR, TW, SPLIT, true, false, false, SPLIT_OUT>(a, o, o, rem0, lds);
        else if (a.nt & 2) col2_tile<T, A, TR, TW, SPLIT, false, false, !SPLIT_OUT, SPLIT_OUT>(a, o, o, rem0, lds);
        else col2_tile<T, A, TR, TW, SPLIT, false, false, false, SPLIT_OUT>(a, o, o, rem0, lds);
    } else {
        col2_tile<T, A, TR, TW, SPLIT, false, false, false, SPLIT_OUT>(a, o, o, rem0, lds);
    }
}

// Split-complex fp32 planes as PLAIN launches (second batch of round 4): a 512-thread work-group runs the two sibling 16-column tiles
// interleaved at lane level (col2_tile WIDE), so that a wave instruction touches whole 128-byte lines of a plane -- the 16-column
// tile reads / writes half lines, and below the chain threshold split plans ran 15-25 % behind interleaved ones
// (profiles/r04_aq_split_chain_mode.log).  One tile = 32 adjacent columns; the hints as fft_col2_kernel.
template <int A, bool TR, bool TW, bool SPLIT, bool SPLIT_OUT>
__global__ void __launch_bounds__(512, 2) fft_col2x_kernel(const TileArgs a) {
    using T = float;
    __shared__ __attribute__((aligned(16))) cplx<T> lds[2 * Col2Lds<A, TR, sizeof(cplx<T>)>::ELEMS];
    const long long col0 = (long long)blockIdx.x * 32;
    const long long o = col0 >> a.logMS;
    const long long rem0 = col0 & ((1ll << a.logMS) - 1);
    const int tid = threadIdx.x;
    if constexpr (TR) {
        if (a.nt & 4) col2_tile<T, A, TR, TW, SPLIT, true, false, false, SPLIT_OUT, true>(a, o, o, rem0, lds, TileNoHook(), tid);
        else if (a.nt & 1) col2_tile<T, A, TR, TW, SPLIT, false, true, false, SPLIT_OUT, true>(a, o, o, rem0, lds, TileNoHook(), tid);
        else col2_tile<T, A, TR, TW, SPLIT, false, false, false, SPLIT_OUT, true>(a, o, o, rem0, lds, TileNoHook(), tid);
    } else {
        if (a.nt & 2) col2_tile<T, A, TR, TW, SPLIT, false, false, true, SPLIT_OUT, true>(a, o, o, rem0, lds, TileNoHook(), tid);
        else col2_tile<T, A, TR, TW, SPLIT, false, false, false, SPLIT_OUT, true>(a, o, o, rem0, lds, TileNoHook(), tid);
    }
}

}  // namespace mifft

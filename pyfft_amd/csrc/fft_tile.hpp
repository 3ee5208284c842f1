// Tile FFT kernels for gfx950 (MI355X).
//
// One work-group owns a tile of P = L * W complex points: W independent length-L transforms.
//   COL tile: the transform runs along a strided axis and the W columns are adjacent in memory
//             (one Stockham pass of a long/strided axis; counterpart of pyfft/kernel.mako:805-1047).
//   ROW tile: the transform runs along the contiguous axis, the tile is W consecutive rows
//             (whole LDS-resident transform; counterpart of pyfft/kernel.mako:725-803).
// Data flow: 16-byte coalesced global loads -> LDS -> [radix stage: LDS -> VGPR butterflies -> LDS]* ->
// 16-byte coalesced global stores.  Every thread keeps PPT = P / NT points in registers per stage and
// runs PPT / R radix-R butterflies (R in {2,4,8,16}), with Stockham autosort indexing so the result is
// in natural order without a bit-reversal pass.
//
// The inverse transform is conj -> forward -> conj, folded into the load and the scaled store.
#pragma once
#include "fft_butterfly.hpp"
#ifndef MIFFT_TREE_MIN_L
#define MIFFT_TREE_MIN_L 64
#endif

namespace mifft {

struct TileArgs {
    const void* in0;
    const void* in1;
    void* out0;
    void* out1;
    const void* tw_L;   // L entries   w(L)^k
    const void* tw_lo;  // 2^tw_shift entries w(L*M)^k
    const void* tw_hi;  // (L*M) >> tw_shift entries w(L*M)^(k << tw_shift)
    long long ostride_in;
    long long ostride_out;
    long long total;  // COL: outer * M * S columns;  ROW: number of rows
    int logMS;        // COL: log2(M*S)
    int logS;         // COL: log2(S)
    int tw_shift;
    int split;      // input side: 1 = two scalar planes (in0 = re, in1 = im), 0 = interleaved in in0
    int split_out;  // output side likewise (differs from `split` only for a plan's internal temp buffer)
    int inverse;
    int has_tw;  // COL: M > 1 -> multiply by the inter-pass twiddle w(L*M)^(l*q)
    int nt;      // bit 0: non-temporal loads of the input, bit 1: non-temporal stores of the output (kernels may ignore)
    double scale;
};

template <int... Rs> struct RadixList {};

// NT: non-temporal access (MIFFT_FLAG_STREAM_SRC / _DST: data read once / not re-read by the plan), interleaved only
template <typename T, bool NT = false> __device__ __forceinline__ void load_pair(const TileArgs& a, long long g, cplx<T>& p0,
                                                                                 cplx<T>& p1) {
    using V4 = T __attribute__((ext_vector_type(4)));
    if (!a.split) {
        const V4* q = reinterpret_cast<const V4*>(reinterpret_cast<const cplx<T>*>(a.in0) + g);
        V4 t;
        if constexpr (NT) t = __builtin_nontemporal_load(q);
        else t = *q;
        p0.x = t.x; p0.y = t.y; p1.x = t.z; p1.y = t.w;
    } else {
        cplx<T> re = *reinterpret_cast<const cplx<T>*>(reinterpret_cast<const T*>(a.in0) + g);
        cplx<T> im = *reinterpret_cast<const cplx<T>*>(reinterpret_cast<const T*>(a.in1) + g);
        p0.x = re.x; p0.y = im.x; p1.x = re.y; p1.y = im.y;
    }
}

// NT: 0 plain, 1 non-temporal (interleaved side only), 2 write-through
template <typename T, int NT = 0> __device__ __forceinline__ void store_pair(const TileArgs& a, long long g, cplx<T> p0,
                                                                                 cplx<T> p1) {
    using V4 = T __attribute__((ext_vector_type(4)));
    if (!a.split_out) {
        V4 t;
        t.x = p0.x; t.y = p0.y; t.z = p1.x; t.w = p1.y;
        V4* q = reinterpret_cast<V4*>(reinterpret_cast<cplx<T>*>(a.out0) + g);
        if constexpr (NT == 2) store_vec_wt(q, t);
        else if constexpr (NT == 1) __builtin_nontemporal_store(t, q);
        else *q = t;
    } else {
        cplx<T> re, im;
        re.x = p0.x; re.y = p1.x; im.x = p0.y; im.y = p1.y;
        if constexpr (NT == 2) {
            store_vec_wt(reinterpret_cast<cplx<T>*>(reinterpret_cast<T*>(a.out0) + g), re);
            store_vec_wt(reinterpret_cast<cplx<T>*>(reinterpret_cast<T*>(a.out1) + g), im);
        } else {
            *reinterpret_cast<cplx<T>*>(reinterpret_cast<T*>(a.out0) + g) = re;
            *reinterpret_cast<cplx<T>*>(reinterpret_cast<T*>(a.out1) + g) = im;
        }
    }
}

// V consecutive points (V = 2: one 16-byte access interleaved / one 8-byte access per plane when split;
// V = 4: split planes only, one 16-byte access per plane)
template <typename T, int V, bool NT = false> __device__ __forceinline__ void load_vec(const TileArgs& a, long long g, cplx<T>* p) {
    if constexpr (V == 2) {
        load_pair<T, NT>(a, g, p[0], p[1]);
    } else {
        using V4 = T __attribute__((ext_vector_type(4)));
        const V4 re = *reinterpret_cast<const V4*>(reinterpret_cast<const T*>(a.in0) + g);
        const V4 im = *reinterpret_cast<const V4*>(reinterpret_cast<const T*>(a.in1) + g);
        p[0].x = re.x; p[0].y = im.x; p[1].x = re.y; p[1].y = im.y;
        p[2].x = re.z; p[2].y = im.z; p[3].x = re.w; p[3].y = im.w;
    }
}
template <typename T, int V, int NT = 0> __device__ __forceinline__ void store_vec(const TileArgs& a, long long g, const cplx<T>* p) {
    if constexpr (V == 1) {
        // one 16-byte point per lane, interleaved output, write-through: the fp64 form of the small-launch store policy.  (A lane's PAIR
        // of fp64 points is 32 bytes = two 16-byte store instructions that each cover every other 16 bytes of a line; written through,
        // every line then reaches the memory side as interleaved fragments: fp64 N = 64 at the reference's 32 MiB 0.50 of the roofline
        // against 0.65 with non-temporal and 0.56 with plain stores -- profiles/r05_fp64_write_through_rows.log.)
        static_assert(NT == 2 && sizeof(cplx<T>) == 16, "single points: fp64 write-through only");
        store_vec_wt(reinterpret_cast<cplx<T>*>(a.out0) + g, p[0]);
    } else if constexpr (V == 2) {
        store_pair<T, NT>(a, g, p[0], p[1]);
    } else {
        using V4 = T __attribute__((ext_vector_type(4)));
        V4 re, im;
        re.x = p[0].x; im.x = p[0].y; re.y = p[1].x; im.y = p[1].y;
        re.z = p[2].x; im.z = p[2].y; re.w = p[3].x; im.w = p[3].y;
        if constexpr (NT == 2) {      // write-through (small launches)
            store_vec_wt(reinterpret_cast<V4*>(reinterpret_cast<T*>(a.out0) + g), re);
            store_vec_wt(reinterpret_cast<V4*>(reinterpret_cast<T*>(a.out1) + g), im);
        } else {
            *reinterpret_cast<V4*>(reinterpret_cast<T*>(a.out0) + g) = re;
            *reinterpret_cast<V4*>(reinterpret_cast<T*>(a.out1) + g) = im;
        }
    }
}

// LDS addressing ---------------------------------------------------------------------------------------
// COL layout A: [idx][c]            (c fastest: the W adjacent columns)
// COL layout B: [c][idx], row pitch L+1 (used for the last stage of a transposing pass)
// ROW layout  : [c][idx + idx/16], row pitch LP
template <int L> struct RowPitch {
    static constexpr int value = L + (L >= 16 ? L / 16 : 1);
};

template <int L, int W, bool ROW> __device__ __forceinline__ int lds_addr(int idx, int c) {
    if constexpr (ROW)
        return c * RowPitch<L>::value + idx + (idx >> 4);
    else
        return idx * W + c;
}

template <typename T, int L, int W, int NT, bool ROW, bool TR, int Ns, typename RL> struct Stages;

template <typename T, int L, int W, int NT, bool ROW, bool TR, int Ns> struct Stages<T, L, W, NT, ROW, TR, Ns, RadixList<>> {
    static __device__ __forceinline__ void run(cplx<T>*, cplx<T>*, const TileArgs&, int, long long) {}
};

template <typename T, int L, int W, int NT, bool ROW, bool TR, int Ns, int R, int... Rest>
struct Stages<T, L, W, NT, ROW, TR, Ns, RadixList<R, Rest...>> {
    static constexpr int P = L * W;
    static constexpr int PPT = P / NT;
    static constexpr int NB = PPT / R;  // butterflies per thread
    static constexpr int LR = L / R;    // butterflies per transform
    static constexpr bool LAST = sizeof...(Rest) == 0;
    static_assert(NB >= 1 && NB * R == PPT, "radix must divide the points per thread");
    static_assert(LR >= 1 && LR * R == L, "radix must divide L");

    static __device__ __forceinline__ void split_bid(int bid, int& j, int& c) {
        if constexpr (ROW) {
            j = bid % LR;
            c = bid / LR;
        } else {
            c = bid % W;
            j = bid / W;
        }
    }

    static __device__ __forceinline__ void run(cplx<T>* lds, cplx<T>* v, const TileArgs& a, int tid, long long col0) {
        const cplx<T>* twL = reinterpret_cast<const cplx<T>*>(a.tw_L);
        // ---- LDS -> registers, stage twiddle, butterfly
        static_for<NB>([&](auto bb) {
            constexpr int b = bb;
            int j, c;
            split_bid(b * NT + tid, j, c);
            static_for<R>([&](auto kk) {
                constexpr int k = kk;
                v[b * R + k] = lds[lds_addr<L, W, ROW>(j + k * LR, c)];
            });
            if constexpr (Ns > 1) {
                const int ai = (j & (Ns - 1)) * (L / (Ns * R));
                if constexpr (L >= MIFFT_TREE_MIN_L && R >= 4) {
                    // w(L)^(k*ai), k < R: R-1 look-ups scattered over the table compete with the data stream for the
                    // 32 KiB L1 (at L = 4096 the table IS 32 KiB: 52 % -> 63 % of roofline; still 68 % -> 71.5 % at
                    // L = 1024), so ONE look-up (index < L/R) and the other powers by a tree of depth <= 4
                    cplx<T> t[R];
                    t[1] = twL[ai];
                    static_for<R - 2>([&](auto kk) {
                        constexpr int k = kk + 2;  // t[k] = t[k/2]^2 (k even) or t[k-1] * t[1]
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
                        v[b * R + k] = cmul<T>(v[b * R + k], twL[k * ai]);
                    });
                }
            }
            Dft<R, T>::run(v + b * R);
        });
        __syncthreads();
        // ---- registers -> LDS at the autosort position
        static_for<NB>([&](auto bb) {
            constexpr int b = bb;
            int j, c;
            split_bid(b * NT + tid, j, c);
            const int idxD = (j & ~(Ns - 1)) * R + (j & (Ns - 1));
            if constexpr (LAST && !ROW) {
                if (a.has_tw) {
                    const cplx<T>* twlo = reinterpret_cast<const cplx<T>*>(a.tw_lo);
                    const cplx<T>* twhi = reinterpret_cast<const cplx<T>*>(a.tw_hi);
                    const unsigned rem = (unsigned)((col0 + c) & ((1ll << a.logMS) - 1));
                    const unsigned l = rem >> a.logS;
                    const unsigned lomask = (1u << a.tw_shift) - 1u;
                    static_for<R>([&](auto kk) {
                        constexpr int k = kk;
                        const unsigned e = l * (unsigned)(idxD + k * Ns);
                        cplx<T> w = cmul<T>(twlo[e & lomask], twhi[e >> a.tw_shift]);
                        v[b * R + k] = cmul<T>(v[b * R + k], w);
                    });
                }
            }
            static_for<R>([&](auto kk) {
                constexpr int k = kk;
                const int idx = idxD + k * Ns;
                if constexpr (LAST && TR)
                    lds[c * (L + 1) + idx] = v[b * R + k];
                else
                    lds[lds_addr<L, W, ROW>(idx, c)] = v[b * R + k];
            });
        });
        __syncthreads();
        if constexpr (!LAST) Stages<T, L, W, NT, ROW, TR, Ns * R, RadixList<Rest...>>::run(lds, v, a, tid, col0);
    }
};

template <typename T, int L, int W, bool ROW, bool TR> struct LdsSize {
    static constexpr int P = L * W;
    static constexpr int value = ROW ? W * RowPitch<L>::value : (TR ? (W * (L + 1) > P ? W * (L + 1) : P) : P);
};

// ---------------------------------------------------------------------------------------------------
template <typename T, int L, int W, int NT, bool ROW, bool TR, typename RL>
__global__ void __launch_bounds__(NT) fft_tile_kernel(const TileArgs a) {
    constexpr int P = L * W;
    constexpr int PPT = P / NT;
    static_assert(PPT * NT == P && PPT >= 2 && (PPT % 2) == 0, "bad tile configuration");
    static_assert(!(ROW && TR), "TR applies to COL tiles only");
    __shared__ __attribute__((aligned(16))) cplx<T> lds[LdsSize<T, L, W, ROW, TR>::value];

    const int tid = threadIdx.x;
    const long long col0 = (long long)blockIdx.x * W;  // first column (COL) / first row (ROW) of the tile
    const long long MSmask = (1ll << a.logMS) - 1;
    cplx<T> v[PPT];

    // Split planes move 4 points (16 bytes per plane) per thread and step when the 4 points are contiguous in
    // memory on both sides; everything else moves 2 points per step.
    constexpr bool kQuadShape = (PPT % 4 == 0) && (ROW ? (L % 4 == 0) : (W % 4 == 0 && L % 4 == 0));
    const bool quad_ok = kQuadShape && (((a.ostride_in | a.ostride_out) & 3) == 0) &&
                         (ROW || (a.logMS >= 2 && (TR || a.logS >= 2)));
    const bool quad_in = quad_ok && a.split, quad_out = quad_ok && a.split_out;
    const T csign = a.inverse ? (T)-1 : (T)1;

    // ---- global -> registers (all loads in flight), then -> LDS
    auto load_phase = [&](auto vv, auto ntc) {
        constexpr int V = vv;
        constexpr bool NTL = (int)ntc != 0;
        static_for<PPT / V>([&](auto ii) {
            constexpr int it = ii;
            const int e = (it * NT + tid) * V;
            cplx<T> p[V];
            static_for<V>([&](auto k) { p[k].x = 0; p[k].y = 0; });
            if constexpr (ROW) {
                const int c = e / L, r = e % L;
                const long long rr = col0 + c;
                if (rr < a.total) load_vec<T, V, NTL>(a, rr * a.ostride_in + r, p);
            } else {
                const int r = e / W, c = e % W;
                const long long cc = col0 + c;
                if (cc < a.total) {
                    const long long o = cc >> a.logMS, rem = cc & MSmask;
                    load_vec<T, V, NTL>(a, o * a.ostride_in + ((long long)r << a.logMS) + rem, p);
                }
            }
            static_for<V>([&](auto k) { v[V * it + k] = p[k]; });
        });
        static_for<PPT / V>([&](auto ii) {
            constexpr int it = ii;
            const int e = (it * NT + tid) * V;
            static_for<V>([&](auto kk) {
                constexpr int k = kk;
                cplx<T> q = v[V * it + k];
                q.y *= csign;
                if constexpr (ROW) {
                    const int c = e / L, r = e % L;
                    lds[lds_addr<L, W, ROW>(r + k, c)] = q;
                } else {
                    const int r = e / W, c = e % W;
                    lds[lds_addr<L, W, ROW>(r, c + k)] = q;
                }
            });
        });
    };
    // (streaming hint only on the plain interleaved path)
    if constexpr (kQuadShape) {
        if (quad_in) load_phase(IC<4>{}, IC<0>{});
        else if (a.nt & 1) load_phase(IC<2>{}, IC<1>{});
        else load_phase(IC<2>{}, IC<0>{});
    } else {
        if (a.nt & 1) load_phase(IC<2>{}, IC<1>{});
        else load_phase(IC<2>{}, IC<0>{});
    }
    __syncthreads();

    Stages<T, L, W, NT, ROW, TR, 1, RL>::run(lds, v, a, tid, col0);

    // ---- LDS -> global (scaled, conjugated back for the inverse)
    const T sx = (T)a.scale;
    const T sy = a.inverse ? -sx : sx;
    auto store_phase = [&](auto vv, auto ntc) {
        constexpr int V = vv;
        constexpr int NTS = (int)ntc;
        static_for<PPT / V>([&](auto ii) {
            constexpr int it = ii;
            const int e = (it * NT + tid) * V;
            cplx<T> p[V];
            long long g;
            bool valid;
            if constexpr (ROW) {
                const int c = e / L, r = e % L;
                const long long rr = col0 + c;
                valid = rr < a.total;
                g = rr * a.ostride_out + r;
                static_for<V>([&](auto k) { p[k] = lds[lds_addr<L, W, ROW>(r + k, c)]; });
            } else if constexpr (TR) {
                // S == 1: out[o][l][q], q contiguous
                const int c = e / L, q = e % L;
                const long long cc = col0 + c;
                valid = cc < a.total;
                const long long o = cc >> a.logMS, rem = cc & MSmask;
                g = o * a.ostride_out + rem * L + q;
                static_for<V>([&](auto k) { p[k] = lds[c * (L + 1) + q + k]; });
            } else {
                // out[o][l][q][j'], j' contiguous (S >= V)
                const int q = e / W, c = e % W;
                const long long cc = col0 + c;
                valid = cc < a.total;
                const long long o = cc >> a.logMS, rem = cc & MSmask;
                const long long l = rem >> a.logS, jp = rem & ((1ll << a.logS) - 1);
                g = o * a.ostride_out + (((l * L) + q) << a.logS) + jp;
                static_for<V>([&](auto k) { p[k] = lds[lds_addr<L, W, ROW>(q, c + k)]; });
            }
            static_for<V>([&](auto k) { p[k].x *= sx; p[k].y *= sy; });
            if (valid) store_vec<T, V, NTS>(a, g, p);
        });
    };
    // fp64, interleaved output, write-through: point by point, so that a store instruction covers whole lines (store_vec, V == 1)
    constexpr bool kPointWise = sizeof(cplx<T>) == 16;
    if constexpr (kQuadShape) {
        // (planes in small launches: write-through 16-byte stores, second batch of round 4; fp64 planes: PAIRS -- four doubles are 32 bytes
        // per lane and plane, the fragmenting form again: fp64 planes N = 512 at 32 MiB 0.47 written through in fours, 0.51 plain)
        if (quad_out && (a.nt & 4) && !kPointWise) store_phase(IC<4>{}, IC<2>{});
        else if (quad_out && !(a.nt & 4)) store_phase(IC<4>{}, IC<0>{});
        else if ((a.nt & 4) && kPointWise && !a.split_out) store_phase(IC<kPointWise ? 1 : 2>{}, IC<2>{});
        else if (a.nt & 4) store_phase(IC<2>{}, IC<2>{});
        else if (a.nt & 2) store_phase(IC<2>{}, IC<1>{});
        else store_phase(IC<2>{}, IC<0>{});
    } else {
        if ((a.nt & 4) && kPointWise && !a.split_out) store_phase(IC<kPointWise ? 1 : 2>{}, IC<2>{});
        else if (a.nt & 4) store_phase(IC<2>{}, IC<2>{});
        else if (a.nt & 2) store_phase(IC<2>{}, IC<1>{});
        else store_phase(IC<2>{}, IC<0>{});
    }
}

}  // namespace mifft

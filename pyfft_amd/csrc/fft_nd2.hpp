// Whole 2-D / 3-D transforms of one fixed shape, register-edged: the compile-time counterpart of fft_nd.hpp for the
// common shapes (a tile of P points = P / (LX*LY*LZ) complete transforms per work-group).
// Same data flow as fft_row2.hpp -- the points live in registers and cross LDS only to be re-distributed between two
// radix stages, optionally as real parts then imaginary parts (HALF: 128 x 128 in fp32 takes 64 KiB instead of 128, two
// work-groups per CU); the last stage stores straight to HBM; the first stage loads straight from HBM when its
// butterflies read runs of >= 128 bytes (EDGE_IN), otherwise after a 16-byte linear load and one more exchange.
// Every stage is described at compile time by (axis length LA, element stride SA, radix R, product Ns of the axis'
// earlier radices), so that an LDS address is one per-thread base plus an immediate offset.  (With run-time geometry
// the same structure needs ~190 VGPRs for the big tile and cannot keep two work-groups per CU; and even at equal
// occupancy the fixed-shape form measures 46 % -> 58 % of roofline at 128 x 128.)
//
// Stage algebra = fft_nd.hpp / fft_tile.hpp (Stockham autosort): butterfly (o, jb, j) of an axis reads
// base + (jb + k*LA/R)*SA, multiplies by w(LA)^(k * (jb mod Ns) * LA/(Ns*R)) and writes base + (idxD + k*Ns)*SA with
// base = o*LA*SA + j, idxD = (jb & ~(Ns-1))*R + (jb & (Ns-1)).
#pragma once
#include "fft_row2.hpp"

namespace mifft {

// TS: the axis' twiddle table belongs to a transform TS times as long (fft_nd2z.hpp: half-length stages on the full axis' table)
template <int AX_, int LA_, int SA_, int Ns_, int R_, int TS_ = 1> struct Nd2StageDesc {
    static constexpr int AX = AX_, LA = LA_, SA = SA_, Ns = Ns_, R = R_, TS = TS_;
};
template <typename... S> struct Nd2StageList {};

// stage descriptors of one axis from its radix list
template <int AX, int LA, int SA, int Ns, typename RL, typename Acc, int TS = 1> struct Nd2AxisStages;
template <int AX, int LA, int SA, int Ns, typename... Acc, int TS>
struct Nd2AxisStages<AX, LA, SA, Ns, RadixList<>, Nd2StageList<Acc...>, TS> {
    using type = Nd2StageList<Acc...>;
};
template <int AX, int LA, int SA, int Ns, int R, int... Rest, typename... Acc, int TS>
struct Nd2AxisStages<AX, LA, SA, Ns, RadixList<R, Rest...>, Nd2StageList<Acc...>, TS> {
    using type = typename Nd2AxisStages<AX, LA, SA, Ns * R, RadixList<Rest...>,
                                        Nd2StageList<Acc..., Nd2StageDesc<AX, LA, SA, Ns, R, TS>>, TS>::type;
};

template <typename T, int P, int NT, bool HALF, typename D> struct Nd2Stage {
    static constexpr int AX = D::AX, LA = D::LA, SA = D::SA, Ns = D::Ns, R = D::R;
    static constexpr int PPT = P / NT, NB = PPT / R, LR = LA / R;
    static_assert(NB >= 1 && NB * R == PPT, "radix must divide the points per thread");
    // pad(base + c) == pad(base) + c + (c >> 4) for the offsets c used below: everything is a power of two, the
    // per-thread part is smaller than the offset unit or a multiple of 16 above it (see fft_row2.hpp)
    using LdsT = typename std::conditional<HALF, T, cplx<T>>::type;

    static __device__ __forceinline__ void geom(int b, int tid, int& base, int& jb) {
        const int bid = b * NT + tid;
        const int j = bid % SA, t = bid / SA;
        jb = t % LR;
        base = (t / LR) * (LA * SA) + j;
    }
    static __device__ __forceinline__ int idxd(int jb) { return (jb & ~(Ns - 1)) * R + (jb & (Ns - 1)); }

    template <int SEL> static __device__ __forceinline__ void fetch(const LdsT* lds, cplx<T>* v, int tid) {
        static_for<NB>([&](auto bb) {
            constexpr int b = bb;
            int base, jb;
            geom(b, tid, base, jb);
            const LdsT* p = lds + row2_pad(base + jb * SA);
            static_for<R>([&](auto kk) {
                constexpr int k = kk;
                constexpr int off = k * LR * SA + ((k * LR * SA) >> 4);
                if constexpr (SEL == 0) v[b * R + k] = p[off];
                else if constexpr (SEL == 1) v[b * R + k].x = p[off];
                else v[b * R + k].y = p[off];
            });
        });
    }
    template <int SEL> static __device__ __forceinline__ void spill(LdsT* lds, const cplx<T>* v, int tid) {
        static_for<NB>([&](auto bb) {
            constexpr int b = bb;
            int base, jb;
            geom(b, tid, base, jb);
            LdsT* p = lds + row2_pad(base + idxd(jb) * SA);
            static_for<R>([&](auto kk) {
                constexpr int k = kk;
                constexpr int off = k * Ns * SA + ((k * Ns * SA) >> 4);
                if constexpr (SEL == 0) p[off] = v[b * R + k];
                else if constexpr (SEL == 1) p[off] = v[b * R + k].x;
                else p[off] = v[b * R + k].y;
            });
        });
    }
    // operands straight from HBM: `inb` = first byte of the transform (wave-uniform)
    // `left` = points from the start of the tile to the end of the data (whole transforms are in or out of range)
    template <bool NTL = false>
    static __device__ __forceinline__ void load(const char* inb, cplx<T>* v, int tid, long long left) {
        static_for<NB>([&](auto bb) {
            constexpr int b = bb;
            int base, jb;
            geom(b, tid, base, jb);
            const unsigned voff = (unsigned)(base + jb * SA) * (unsigned)sizeof(cplx<T>);
            const bool ok = base < left;
            static_for<R>([&](auto kk) {
                constexpr int k = kk;
                cplx<T> p;
                p.x = 0; p.y = 0;
                if (ok) {
                    const cplx<T>* q = reinterpret_cast<const cplx<T>*>(inb + (size_t)(k * LR * SA) * sizeof(cplx<T>) + voff);
                    if constexpr (NTL) p = __builtin_nontemporal_load(q);
                    else p = *q;
                }
                v[b * R + k] = p;
            });
        });
    }
    // NTS: 0 plain, 1 non-temporal, 2 write-through (sc1) stores
    template <int NTS = 0>
    static __device__ __forceinline__ void store(char* outb, const cplx<T>* v, int tid, T sx, T sy, long long left) {
        static_for<NB>([&](auto bb) {
            constexpr int b = bb;
            int base, jb;
            geom(b, tid, base, jb);
            const unsigned voff = (unsigned)(base + idxd(jb) * SA) * (unsigned)sizeof(cplx<T>);
            if (base < left) {
                static_for<R>([&](auto kk) {
                    constexpr int k = kk;
                    cplx<T> p = v[b * R + k];
                    p.x *= sx;
                    p.y *= sy;
                    char* kb = outb + (size_t)(k * Ns * SA) * sizeof(cplx<T>);
                    cplx<T>* q = reinterpret_cast<cplx<T>*>(kb + voff);
                    if constexpr (NTS == 2) store_wt<T>(kb, voff, p);
                    else if constexpr (NTS == 1) __builtin_nontemporal_store(p, q);
                    else *q = p;
                });
            }
        });
    }
    static __device__ __forceinline__ void compute(cplx<T>* v, const cplx<T>* tw, int tid) {
        static_for<NB>([&](auto bb) {
            constexpr int b = bb;
            if constexpr (Ns > 1) {
                int base, jb;
                geom(b, tid, base, jb);
                const int ai = (jb & (Ns - 1)) * (LA / (Ns * R));
                row2_twiddle<T, R, D::TS>(tw, ai, v + b * R);
            }
            Dft<R, T>::run(v + b * R);
        });
    }
};

template <typename T, int P, int NT, bool HALF, bool FIRST, typename SL> struct Nd2Chain;

template <typename T, int P, int NT, bool HALF, bool FIRST, typename D, typename... Rest>
struct Nd2Chain<T, P, NT, HALF, FIRST, Nd2StageList<D, Rest...>> {
    using St = Nd2Stage<T, P, NT, HALF, D>;
    using LdsT = typename St::LdsT;
    static constexpr bool LAST = sizeof...(Rest) == 0;

    // v holds the operands of this stage; tw[ax] = twiddle table of axis ax
    static __device__ __forceinline__ void run(LdsT* lds, cplx<T>* v, const cplx<T>* const* tw, int tid, char* outb, T sx,
                                               T sy, long long left, int nt_out) {
        St::compute(v, tw[D::AX], tid);
        if constexpr (LAST) {
            if (nt_out == 2) St::template store<2>(outb, v, tid, sx, sy, left);
            else if (nt_out == 1) St::template store<1>(outb, v, tid, sx, sy, left);  // MIFFT_FLAG_STREAM_DST
            else St::template store<0>(outb, v, tid, sx, sy, left);
        } else {
            using NextChain = Nd2Chain<T, P, NT, HALF, false, Nd2StageList<Rest...>>;
            using Next = typename NextChain::St;
            if constexpr (!FIRST) __syncthreads();  // everybody has fetched its operands of this stage
            if constexpr (!HALF) {
                St::template spill<0>(lds, v, tid);
                __syncthreads();
                Next::template fetch<0>(lds, v, tid);
            } else {
                St::template spill<1>(lds, v, tid);
                __syncthreads();
                Next::template fetch<1>(lds, v, tid);  // the real-part registers are free again: reuse them
                __syncthreads();
                St::template spill<2>(lds, v, tid);
                __syncthreads();
                Next::template fetch<2>(lds, v, tid);
            }
            NextChain::run(lds, v, tw, tid, outb, sx, sy, left, nt_out);
        }
    }
};

// The same chain with the last stage's results handed to `sink(St{}, v)` instead of the dense store (fft_colx.hpp: strided
// and transposing stores with an inter-pass twiddle).
template <typename T, int P, int NT, bool HALF, bool FIRST, typename SL> struct Nd2ChainSink;
template <typename T, int P, int NT, bool HALF, bool FIRST, typename D, typename... Rest>
struct Nd2ChainSink<T, P, NT, HALF, FIRST, Nd2StageList<D, Rest...>> {
    using St = Nd2Stage<T, P, NT, HALF, D>;
    using LdsT = typename St::LdsT;
    template <typename Sink>
    static __device__ __forceinline__ void run(LdsT* lds, cplx<T>* v, const cplx<T>* const* tw, int tid, Sink& sink) {
        St::compute(v, tw[D::AX], tid);
        if constexpr (sizeof...(Rest) == 0) {
            sink(St{}, v);
        } else {
            using NextChain = Nd2ChainSink<T, P, NT, HALF, false, Nd2StageList<Rest...>>;
            using Next = typename NextChain::St;
            if constexpr (!FIRST) __syncthreads();
            if constexpr (!HALF) {
                St::template spill<0>(lds, v, tid);
                __syncthreads();
                Next::template fetch<0>(lds, v, tid);
            } else {
                St::template spill<1>(lds, v, tid);
                __syncthreads();
                Next::template fetch<1>(lds, v, tid);
                __syncthreads();
                St::template spill<2>(lds, v, tid);
                __syncthreads();
                Next::template fetch<2>(lds, v, tid);
            }
            // the next stage gets an opaque copy of the thread index: what it derives from it (LDS / global addresses, twiddle
            // look-ups) is then computed when that stage runs instead of being hoisted above the first stage's loads, where it
            // pushed 36-80 registers of freshly loaded operands into scratch
            int tnext = tid;
            asm volatile("" : "+v"(tnext));
            NextChain::run(lds, v, tw, tnext, sink);
        }
    }
};
template <typename T, int P, int NT, bool HALF, bool FIRST, typename SL, typename LdsT, typename Sink>
__device__ __forceinline__ void nd2_chain_sink(LdsT* lds, cplx<T>* v, const cplx<T>* const* tw, int tid, Sink& sink) {
    Nd2ChainSink<T, P, NT, HALF, FIRST, SL>::run(lds, v, tw, tid, sink);
}

template <typename A, typename B> struct Nd2Concat;
template <typename... A, typename... B> struct Nd2Concat<Nd2StageList<A...>, Nd2StageList<B...>> {
    using type = Nd2StageList<A..., B...>;
};
template <typename SL> struct Nd2First;
template <typename D, typename... Rest> struct Nd2First<Nd2StageList<D, Rest...>> { using type = D; };

// A tile of P points = P / (LX*LY*LZ) whole (LZ, LY, LX) transforms (x contiguous) per work-group.  TileArgs: in0 / out0
// interleaved, total = number of POINTS, tw_L / tw_lo / tw_hi = w(LX) / w(LY) / w(LZ) tables, inverse, scale.
template <typename T, int LX, int LY, int LZ, int P, int NT, bool HALF, int OCC, bool EDGE_IN, typename RLX, typename RLY,
          typename RLZ>
__global__ void __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(OCC))) fft_nd2_kernel(const TileArgs a) {
    constexpr int PPT = P / NT;
    static_assert(PPT * NT == P && P % (LX * LY * LZ) == 0 && PPT % 2 == 0, "bad tile");
    static_assert(EDGE_IN || !HALF, "the linear-load form is built for full-complex exchanges only");
    using SX = typename Nd2AxisStages<0, LX, 1, 1, RLX, Nd2StageList<>>::type;
    using SY = typename Nd2AxisStages<1, LY, LX, 1, RLY, Nd2StageList<>>::type;
    using SZ = typename Nd2AxisStages<2, LZ, LX * LY, 1, RLZ, Nd2StageList<>>::type;
    using SL = typename Nd2Concat<typename Nd2Concat<SX, SY>::type, SZ>::type;
    using First = Nd2Stage<T, P, NT, HALF, typename Nd2First<SL>::type>;
    using LdsT = typename std::conditional<HALF, T, cplx<T>>::type;
    __shared__ __attribute__((aligned(16))) LdsT lds[P + P / 16];
    const int tid = threadIdx.x;
    const long long g0 = (long long)blockIdx.x * P;
    const long long left = a.total - g0;
    const char* inb = reinterpret_cast<const char*>(reinterpret_cast<const cplx<T>*>(a.in0) + g0);
    char* outb = reinterpret_cast<char*>(reinterpret_cast<cplx<T>*>(a.out0) + g0);
    const cplx<T>* tw[3] = {reinterpret_cast<const cplx<T>*>(a.tw_L), reinterpret_cast<const cplx<T>*>(a.tw_lo),
                            reinterpret_cast<const cplx<T>*>(a.tw_hi)};
    const T sx = (T)a.scale;
    const T sy = a.inverse ? -sx : sx;
    cplx<T> v[PPT];
    if constexpr (EDGE_IN) {
        if (a.nt & 1) First::template load<true>(inb, v, tid, left);  // MIFFT_FLAG_STREAM_SRC
        else First::template load<false>(inb, v, tid, left);
    } else {
        // 16-byte linear load (thread: points 2*(it*NT + tid), +1), then one exchange into the first stage's order
        using V4 = T __attribute__((ext_vector_type(4)));
        static_for<PPT / 2>([&](auto ii) {
            constexpr int it = ii;
            const unsigned e = (unsigned)(it * NT + tid) * 2u;
            V4 q = {0, 0, 0, 0};
            if ((long long)e < left) q = *reinterpret_cast<const V4*>(inb + e * (unsigned)sizeof(cplx<T>));
            v[2 * it].x = q.x; v[2 * it].y = q.y; v[2 * it + 1].x = q.z; v[2 * it + 1].y = q.w;
        });
        static_for<PPT / 2>([&](auto ii) {
            constexpr int it = ii;
            LdsT* p = lds + row2_pad((it * NT + tid) * 2);
            p[0] = v[2 * it];
            p[1] = v[2 * it + 1];
        });
        __syncthreads();
        First::template fetch<0>(lds, v, tid);
        __syncthreads();
    }
    if (a.inverse) static_for<PPT>([&](auto i) { v[i].y = -v[i].y; });
    Nd2Chain<T, P, NT, HALF, true, SL>::run(lds, v, tw, tid, outb, sx, sy, left, (a.nt & 4) ? 2 : ((a.nt & 2) ? 1 : 0));
}

template <typename T, int LX, int LY, int LZ, int P, int NT, bool HALF, int OCC, bool EDGE_IN, typename RLX, typename RLY,
          typename RLZ = RadixList<>>
static inline int launch_nd2(const TileArgs* a, hipStream_t s) {
    const long long tiles = (a->total + P - 1) / P;
    if (tiles <= 0) return 0;
    if (tiles > 2147483647ll) return -1;
    hipLaunchKernelGGL((fft_nd2_kernel<T, LX, LY, LZ, P, NT, HALF, OCC, EDGE_IN, RLX, RLY, RLZ>), dim3((unsigned)tiles),
                       dim3(NT), 0, s, *a);
    return (int)hipGetLastError();
}

// ---- automatic configuration of a shape --------------------------------------------------------------------------
// Radix list of an axis of length L with radices <= MAXR: full MAXR factors plus one smaller remainder factor, the
// remainder FIRST on the first axis (its first stage then reads the longest runs from HBM), LAST on the others (the last
// stage of all then writes the longest runs).
template <typename A, typename B> struct RadixCat;
template <int... A, int... B> struct RadixCat<RadixList<A...>, RadixList<B...>> { using type = RadixList<A..., B...>; };
template <int L, int MAXR> struct RadixFull {  // L = MAXR^k exactly
    using type = typename RadixCat<RadixList<MAXR>, typename RadixFull<L / MAXR, MAXR>::type>::type;
};
template <int MAXR> struct RadixFull<1, MAXR> { using type = RadixList<>; };
constexpr int nd2_rem(int L, int maxr) {  // the factor of L left over after all full MAXR factors
    while (L >= maxr) L /= maxr;
    return L;
}
template <int L, int MAXR, bool REM_FIRST> struct AutoRadix {
    static constexpr int REM = nd2_rem(L, MAXR);
    using Full = typename RadixFull<L / REM, MAXR>::type;
    using Rem = typename std::conditional<(REM > 1), RadixList<REM>, RadixList<>>::type;
    using type = typename std::conditional<REM_FIRST, typename RadixCat<Rem, Full>::type,
                                           typename RadixCat<Full, Rem>::type>::type;
};
template <typename RL> struct RadixFirst { static constexpr int value = 1; };
template <int R, int... Rest> struct RadixFirst<RadixList<R, Rest...>> { static constexpr int value = R; };

// Tile and launch shape for an (X, Y, Z) transform of type T: the smallest tile of 1 / 2 / 4 / 8 x TILE0 points that holds
// one transform (TILE0 = 4096 points fp32, 2048 fp64: 32 KiB of LDS, four work-groups per CU).  4 x TILE0 ("big") runs in
// half-exchange form with twice the points per thread, so that two work-groups still share a CU; 8 x TILE0 ("huge":
// 128 KiB of LDS even as scalars, one work-group per CU) still beats one HBM round trip per axis.
template <typename T, int X, int Y, int Z> struct Nd2Auto {
    static constexpr int N = X * Y * Z;
    static constexpr bool F32 = sizeof(T) == 4;
    static constexpr int TILE0 = F32 ? 4096 : 2048;
    static constexpr int PPT0 = F32 ? 16 : 8;
    static_assert(N <= 8 * TILE0, "shape too large for one tile");
    static constexpr bool HUGE = N > 4 * TILE0;
    static constexpr bool BIG = !HUGE && N > 2 * TILE0;
    static constexpr int P = N <= TILE0 ? TILE0 : N <= 2 * TILE0 ? 2 * TILE0 : N <= 4 * TILE0 ? 4 * TILE0 : 8 * TILE0;
    static constexpr bool HALF = BIG || HUGE;
    // points per thread: fp32 16 / 32 (big) / 64 (huge, 512 threads, ~240 VGPRs); fp64 8 / 16 (big) / 16 (huge, 1024 threads)
    static constexpr int PPT = HUGE ? (F32 ? 64 : 16) : BIG ? 2 * PPT0 : PPT0;
    static constexpr int NT = P / PPT;
    static constexpr int OCC = HUGE ? (F32 ? 2 : 4) : BIG ? 4 : 1;
    // largest radix: 16 (fp32) / 8 (fp64) -- radix-32 butterflies cost too many registers at 128 VGPRs; the huge tiles
    // have the registers (fp32: 32) or the points per thread (fp64: 16) for one more factor of two per stage
    static constexpr int MAXR = HUGE ? (F32 ? 32 : 16) : PPT0;
    // first axis with more than one point takes the remainder factor first
    static constexpr bool XF = X > 1, YF = !XF && Y > 1;
    using RLX = typename AutoRadix<X, MAXR, true>::type;
    // round 5: on the two-per-CU fp32 tiles (32 points per thread) a 32-point y or z axis is ONE radix-32 stage instead of 16 x 2 -- one
    // LDS exchange less: (16, 32, 32) 0.647 -> 0.710, (32, 16, 32) 0.649 -> 0.714, (8, 32, 64) 0.635 -> 0.698, (32, 512) 0.587 -> 0.641,
    // (32, 32, 16) 0.584 -> 0.624 at 1 GiB (profiles/r05_nd2_radix32_yz_axes_ab.log).  Not the x axis: a radix-32 FIRST stage measured 4-6
    // points slower ((512, 32) 0.628 -> 0.592)
    static constexpr bool R32 = BIG && F32;
    // (the fp64 analogue -- a 16-point y / z axis as one radix-16 stage on the 16-points-per-thread tiles -- measured no gain: (16, 16, 32)
    // 0.711, (32, 16, 16) 0.711 against 0.713 / 0.701: those tiles already run at 0.70)
    using RLY = typename std::conditional<(R32 && Y == 32), RadixList<32>, typename AutoRadix<Y, MAXR, YF>::type>::type;
    using RLZ = typename std::conditional<(R32 && Z == 32), RadixList<32>, typename AutoRadix<Z, MAXR, false>::type>::type;
    // the first stage reads runs of (X / first radix) points: straight from HBM when that is >= 128 bytes
    static constexpr bool EDGE_IN = HALF || (X > 1 && (X / RadixFirst<RLX>::value) * (int)sizeof(cplx<T>) >= 128);
};

template <typename T, int X, int Y, int Z> static inline int launch_nd2_auto(const TileArgs* a, hipStream_t s) {
    using C = Nd2Auto<T, X, Y, Z>;
#ifdef MIFFT_ND2_HUGE_AB
    // round 5 A/B (MIFFT_DEBUG_ALT_ROWS = 4): the one-tile-per-CU fp32 shapes on 1024 threads x 32 points (four waves per SIMD, radices
    // <= 16) instead of 512 threads x 64 points (two waves per SIMD, radix 32)
    if constexpr (C::HUGE && C::F32) {
        if (mifft_debug_get(MIFFT_DEBUG_ALT_ROWS) == 4)
            return launch_nd2<T, X, Y, Z, C::P, 1024, true, 4, true, typename AutoRadix<X, 16, true>::type,
                              typename AutoRadix<Y, 16, C::YF>::type, typename AutoRadix<Z, 16, false>::type>(a, s);
    }
#endif
    return launch_nd2<T, X, Y, Z, C::P, C::NT, C::HALF, C::OCC, C::EDGE_IN, typename C::RLX, typename C::RLY,
                      typename C::RLZ>(a, s);
}

}  // namespace mifft

// "Pass pair" kernels: TWO consecutive Stockham passes of a plan's chain run by one launch, on tiles that hold the points of
// both -- so that a 3-D transform whose (y, x) plane does not fit a work-group still crosses HBM twice instead of three
// times (BASELINE config 4, 256^3 fp64: 3.0 x the algorithmic bytes over L2 <-> fabric with one pass per axis).
//
// The y axis (length ny = R0 * R1) is factored as the chain factors a long axis (pyfft/kernel.py:259-283,
// kernel.mako:805-1047), COL(L = R0, M = R1, S = nx) then COL(L = R1, M = 1, S = nx * R0), and the four passes
//     ROW x | COL y (R0, M = R1) | COL y (R1, M = 1) | COL z
// are run as two launches:
//   XY  = ROW x + COL y (R0):  tile (z, l): the R0 rows y = r * R1 + l of one plane, whole x (R0 * nx points: strided rows in,
//         a contiguous block of R0 rows out -- Stockham autosort puts row l * R0 + q there), inter-pass twiddle w(ny)^(l * q)
//   YZ  = COL y (R1) + COL z:  tile (q, x-chunk): W adjacent x (128-byte segments; 16 x 8 bytes per plane for split fp64 output) of the rows y = r * R0 + q, all z
//         (W * R1 * nz points, every point of a segment gathered / scattered in place)
// Both are the register-edged, LDS-exchanged stage chain of fft_nd2.hpp (same Nd2Stage arithmetic, same exchanges) on a
// tile-LOCAL dense index space (d0, d1, d2); what is new is the map from that space to global memory: every tile dimension
// has its own compile-time global stride on the input and on the output side, and an axis may be carried along untransformed
// (the W adjacent columns of YZ).  All geometry is a template parameter: these kernels exist for a handful of fixed shapes.
#pragma once
#include "fft_nd2.hpp"

namespace mifft {

struct PairArgs {
    const void* in0;     // interleaved data, or the real plane (XY of a split-plane plan)
    const void* in1;     // imaginary plane or null
    void* out0;          // interleaved data, or the real plane (YZ of a split-plane plan)
    void* out1;
    const void* tw[3];   // w(len)^k table of the transformed length of tile dimension 0 / 1 / 2 (null when not transformed)
    const void* tw_lo;   // XY: the inter-pass twiddle of the y axis as the chain's two-level table: w(ny)^e =
    const void* tw_hi;   //     tw_lo[e & (2^tw_shift - 1)] * tw_hi[e >> tw_shift]
    int tw_shift;
    long long tiles;     // number of tiles = grid
    int inverse;
    int nt;              // bit 0: non-temporal loads, bit 1: non-temporal stores, bit 2: write-through (sc1) stores
    double scale;
};

// Tile-local dense space [E2][E1][E0] (d0 fastest) -> global element offsets.  Tile number -> (c0, c1, o) with extents
// (C0, C1, anything): tile base = c0 * B?0 + c1 * B?1 + o * BO on the input (I) / output (O) side.
template <int E0_, int E1_, int E2_, long long GI0_, long long GI1_, long long GI2_, long long GO0_, long long GO1_, long long GO2_,
          int C0_, int C1_, long long BI0_, long long BI1_, long long BO0_, long long BO1_, long long BOUTER_>
struct PairMap {
    static constexpr int E0 = E0_, E1 = E1_, E2 = E2_, C0 = C0_, C1 = C1_;
    static constexpr long long GI[3] = {GI0_, GI1_, GI2_};
    static constexpr long long GO[3] = {GO0_, GO1_, GO2_};
    static constexpr long long BI0 = BI0_, BI1 = BI1_, BO0 = BO0_, BO1 = BO1_, BOUTER = BOUTER_;
    static __device__ __forceinline__ unsigned in_off(int e) {
        return (unsigned)((e % E0) * GI0_ + ((e / E0) % E1) * GI1_ + (e / (E0 * E1)) * GI2_);
    }
    static __device__ __forceinline__ unsigned out_off(int e) {
        return (unsigned)((e % E0) * GO0_ + ((e / E0) % E1) * GO1_ + (e / (E0 * E1)) * GO2_);
    }
};

// first-stage operands straight from HBM through the map.  SPLIT: two scalar planes (inb / inb1 = the tile's first real / imaginary
// scalar), else interleaved complex numbers at inb.
template <typename T, typename St, typename MAP, bool NTL, bool SPLIT>
__device__ __forceinline__ void pair_load(const char* inb, const char* inb1, cplx<T>* v, int tid) {
    constexpr long long G = MAP::GI[St::AX];
    constexpr unsigned ESZ = SPLIT ? sizeof(T) : sizeof(cplx<T>);
    static_for<St::NB>([&](auto bb) {
        constexpr int b = bb;
        int base, jb;
        St::geom(b, tid, base, jb);
        const unsigned voff = MAP::in_off(base + jb * St::SA) * ESZ;
        static_for<St::R>([&](auto kk) {
            constexpr int k = kk;
            const size_t koff = (size_t)((long long)(k * St::LR) * G) * ESZ;
            if constexpr (SPLIT) {
                const T* qr = reinterpret_cast<const T*>(inb + koff + voff);
                const T* qi = reinterpret_cast<const T*>(inb1 + koff + voff);
                if constexpr (NTL) {
                    v[b * St::R + k].x = __builtin_nontemporal_load(qr);
                    v[b * St::R + k].y = __builtin_nontemporal_load(qi);
                } else {
                    v[b * St::R + k].x = *qr;
                    v[b * St::R + k].y = *qi;
                }
            } else {
                const cplx<T>* q = reinterpret_cast<const cplx<T>*>(inb + koff + voff);
                if constexpr (NTL) v[b * St::R + k] = __builtin_nontemporal_load(q);
                else v[b * St::R + k] = *q;
            }
        });
    });
}

// NTS: 0 plain, 1 non-temporal, 2 write-through stores
template <typename T, typename St, typename MAP, int NTS, bool TWOUT, bool SPLIT>
__device__ __forceinline__ void pair_store(char* outb, char* outb1, const cplx<T>* v, int tid, T sx, T sy, const PairArgs& a, int lrow) {
    constexpr long long G = MAP::GO[St::AX];
    constexpr unsigned ESZ = SPLIT ? sizeof(T) : sizeof(cplx<T>);
    static_assert(!TWOUT || St::AX == 1, "the twiddled store belongs to the last stage of tile dimension 1");
    // Inter-pass twiddle w(ny)^(l * q) of the R results q = q0 + k * Ns of a butterfly (round 4): every FOURTH factor is looked up in
    // the two-level table, the three behind it are one multiplication each by the step w(ny)^(l * Ns) -- which is the same for the
    // whole tile (depth <= 3, as the strided kernels of fft_col2.hpp: fp32 error ~2.5e-7 max).  Rounds 2-3 looked every factor up:
    // two table loads and two complex multiplications per element, 32 cached loads per thread in the XY tile of 128^3.
    // Round 6: the anchors are looked up TWO GROUPS AHEAD of their use.  Looked up inside their own group (round 4) the two table loads
    // sat behind the group's predecessor's four stores, and with one in-order memory counter per wave the wait for them was a wait for
    // those write-through stores to be acknowledged -- eight store round trips in a row per XY tile of 128^3 (the ISA read
    // "2 x load, s_waitcnt vmcnt(0), 4 x store" eight times over).  Two groups ahead the wait covers stores that are two groups old.
    const cplx<T>* lo = reinterpret_cast<const cplx<T>*>(a.tw_lo);
    const cplx<T>* hi = reinterpret_cast<const cplx<T>*>(a.tw_hi);
    const unsigned lomask = (1u << a.tw_shift) - 1u;
    auto look = [&](unsigned e) { return cmul<T>(lo[e & lomask], hi[e >> a.tw_shift]); };
    cplx<T> wstep = {(T)1, (T)0};
    if constexpr (TWOUT && St::R > 1) wstep = look((unsigned)(lrow * St::Ns));
    constexpr int GR = (St::R + 3) / 4, NG = St::NB * GR;      // groups of four results per butterfly, groups per thread
    cplx<T> alo[2], ahi[2];
    auto issue = [&](auto gc) {
        constexpr int gi = gc, b = gi / GR, k = (gi % GR) * 4;
        int base, jb;
        St::geom(b, tid, base, jb);
        const int e0 = base + St::idxd(jb) * St::SA;
        const int q0 = (e0 / MAP::E0) % MAP::E1;
        const unsigned e = (unsigned)(lrow * (q0 + k * St::Ns));
        alo[gi % 2] = lo[e & lomask];
        ahi[gi % 2] = hi[e >> a.tw_shift];
    };
    if constexpr (TWOUT) {
        issue(IC<0>{});
        if constexpr (NG > 1) issue(IC<1>{});
        __builtin_amdgcn_sched_barrier(0);
    }
    static_for<St::NB>([&](auto bb) {
        constexpr int b = bb;
        int base, jb;
        St::geom(b, tid, base, jb);
        const int e0 = base + St::idxd(jb) * St::SA;
        const unsigned voff = MAP::out_off(e0) * ESZ;
        cplx<T> wcur = {(T)1, (T)0};
        static_for<St::R>([&](auto kk) {
            constexpr int k = kk;
            cplx<T> p = v[b * St::R + k];
            if constexpr (TWOUT) {
                if constexpr (k % 4 == 0) {
                    constexpr int gi = b * GR + k / 4;
                    wcur = cmul<T>(alo[gi % 2], ahi[gi % 2]);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (gi + 2 < NG) {
                        issue(IC<gi + 2>{});
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
                    wcur = cmul<T>(wcur, wstep);
                }
                p = cmul<T>(p, wcur);
            }
            p.x *= sx;
            p.y *= sy;
            const size_t koff = (size_t)((long long)(k * St::Ns) * G) * ESZ;
            if constexpr (SPLIT) {
                T* qr = reinterpret_cast<T*>(outb + koff + voff);
                T* qi = reinterpret_cast<T*>(outb1 + koff + voff);
                if constexpr (NTS != 0) {
                    __builtin_nontemporal_store(p.x, qr);
                    __builtin_nontemporal_store(p.y, qi);
                } else {
                    *qr = p.x;
                    *qi = p.y;
                }
            } else {
                char* kb = outb + koff;
                cplx<T>* q = reinterpret_cast<cplx<T>*>(kb + voff);
                if constexpr (NTS == 2) store_wt<T>(kb, voff, p);
                else if constexpr (NTS == 1) __builtin_nontemporal_store(p, q);
                else *q = p;
            }
        });
    });
}

template <typename T, int P, int NT, bool HALF, bool FIRST, typename MAP, bool TWOUT, bool SPLIT_OUT, typename SL> struct PairChain;

template <typename T, int P, int NT, bool HALF, bool FIRST, typename MAP, bool TWOUT, bool SPLIT_OUT, typename D, typename... Rest>
struct PairChain<T, P, NT, HALF, FIRST, MAP, TWOUT, SPLIT_OUT, Nd2StageList<D, Rest...>> {
    using St = Nd2Stage<T, P, NT, HALF, D>;
    using LdsT = typename St::LdsT;
    static constexpr bool LAST = sizeof...(Rest) == 0;

    static __device__ __forceinline__ void run(LdsT* lds, cplx<T>* v, const cplx<T>* const* tw, int tid, char* outb, char* outb1,
                                               T sx, T sy, int nt_out, const PairArgs& a, int lrow) {
        St::compute(v, tw[D::AX], tid);
        if constexpr (LAST) {
            if (nt_out == 2) pair_store<T, St, MAP, 2, TWOUT, SPLIT_OUT>(outb, outb1, v, tid, sx, sy, a, lrow);
            else if (nt_out == 1) pair_store<T, St, MAP, 1, TWOUT, SPLIT_OUT>(outb, outb1, v, tid, sx, sy, a, lrow);
            else pair_store<T, St, MAP, 0, TWOUT, SPLIT_OUT>(outb, outb1, v, tid, sx, sy, a, lrow);
        } else {
            using NextChain = PairChain<T, P, NT, HALF, false, MAP, TWOUT, SPLIT_OUT, Nd2StageList<Rest...>>;
            using Next = typename NextChain::St;
            if constexpr (!FIRST) __syncthreads();  // everybody has fetched its operands of this stage
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
            NextChain::run(lds, v, tw, tid, outb, outb1, sx, sy, nt_out, a, lrow);
        }
    }
};

// One tile: `bin` / `bout` = element offset of the tile's first point on the input / output side, `lrow` = the tile's row index l in
// the inter-pass twiddle of an XY tile.  Shared by the plain launch below and the persistent two-pair kernel (fft_fusedp.hpp).
// CFG: P, NT, HALF, OCC, MAP, SL (stage list over the tile-local space), TWOUT, SPLIT_IN, SPLIT_OUT
template <typename T, typename CFG, typename LdsT>
__device__ __forceinline__ void pair_tile(const PairArgs& a, const long long bin, const long long bout, const int lrow, LdsT* lds,
                                          const int tid, const bool nt_in, const int nt_out) {
    using MAP = typename CFG::MAP;
    using SL = typename CFG::SL;
    constexpr int P = CFG::P, NT = CFG::NT, PPT = P / NT;
    constexpr bool HALF = CFG::HALF;
    static_assert(P == MAP::E0 * MAP::E1 * MAP::E2 && PPT * NT == P, "bad tile");
    using First = Nd2Stage<T, P, NT, HALF, typename Nd2First<SL>::type>;
    constexpr bool SI = CFG::SPLIT_IN, SO = CFG::SPLIT_OUT;
    const char* inb = SI ? reinterpret_cast<const char*>(reinterpret_cast<const T*>(a.in0) + bin)
                         : reinterpret_cast<const char*>(reinterpret_cast<const cplx<T>*>(a.in0) + bin);
    const char* inb1 = SI ? reinterpret_cast<const char*>(reinterpret_cast<const T*>(a.in1) + bin) : nullptr;
    char* outb = SO ? reinterpret_cast<char*>(reinterpret_cast<T*>(a.out0) + bout)
                    : reinterpret_cast<char*>(reinterpret_cast<cplx<T>*>(a.out0) + bout);
    char* outb1 = SO ? reinterpret_cast<char*>(reinterpret_cast<T*>(a.out1) + bout) : nullptr;
    const cplx<T>* tw[3] = {reinterpret_cast<const cplx<T>*>(a.tw[0]), reinterpret_cast<const cplx<T>*>(a.tw[1]),
                            reinterpret_cast<const cplx<T>*>(a.tw[2])};
    const T sx = (T)a.scale;
    const T sy = a.inverse ? -sx : sx;
    cplx<T> v[PPT];
    if (nt_in) pair_load<T, First, MAP, true, SI>(inb, inb1, v, tid);
    else pair_load<T, First, MAP, false, SI>(inb, inb1, v, tid);
    if (a.inverse) static_for<PPT>([&](auto i) { v[i].y = -v[i].y; });
    PairChain<T, P, NT, HALF, true, MAP, CFG::TWOUT, SO, SL>::run(lds, v, tw, tid, outb, outb1, sx, sy, nt_out, a, lrow);
}

template <typename T, typename CFG>
__global__ void __launch_bounds__(CFG::NT) __attribute__((amdgpu_waves_per_eu(CFG::OCC))) fft_pair_kernel(const PairArgs a) {
    using MAP = typename CFG::MAP;
    constexpr int P = CFG::P;
    using LdsT = typename std::conditional<CFG::HALF, T, cplx<T>>::type;
    __shared__ __attribute__((aligned(16))) LdsT lds[P + P / 16];
    const unsigned tile = blockIdx.x;
    const unsigned c0 = tile % (unsigned)MAP::C0, c1 = (tile / (unsigned)MAP::C0) % (unsigned)MAP::C1;
    const long long o = tile / (unsigned)(MAP::C0 * MAP::C1);
    const long long bin = o * MAP::BOUTER + (long long)c0 * MAP::BI0 + (long long)c1 * MAP::BI1;
    const long long bout = o * MAP::BOUTER + (long long)c0 * MAP::BO0 + (long long)c1 * MAP::BO1;
    pair_tile<T, CFG>(a, bin, bout, (int)c0, lds, (int)threadIdx.x, (a.nt & 1) != 0, (a.nt & 4) ? 2 : ((a.nt & 2) ? 1 : 0));
}

// ---- the two tile kinds for a (NZ, NY, NX) transform with NY = R0 * R1 ------------------------------------------------
// XY: tile dims (x: NX, r: R0); tile coordinates c0 = l < R1, o = plane (z and batch)
template <typename T, int NX, int R0, int R1, int NT_, bool HALF_, int OCC_, typename RLX, typename RLY, bool SPLIT_IN_ = false>
struct PairXY {
    static constexpr int NY = R0 * R1;
    static constexpr int P = NX * R0, NT = NT_, OCC = OCC_;
    static constexpr bool HALF = HALF_, TWOUT = true, SPLIT_IN = SPLIT_IN_, SPLIT_OUT = false;
    using MAP = PairMap<NX, R0, 1, /*GI*/ 1, (long long)R1 * NX, 0, /*GO*/ 1, NX, 0, /*C*/ R1, 1,
                        /*BI*/ NX, 0, /*BO*/ (long long)R0 * NX, 0, /*outer: one plane*/ (long long)NX * NY>;
    using SX = typename Nd2AxisStages<0, NX, 1, 1, RLX, Nd2StageList<>>::type;
    using SY = typename Nd2AxisStages<1, R0, NX, 1, RLY, Nd2StageList<>>::type;
    using SL = typename Nd2Concat<SX, SY>::type;
};

// YZ: tile dims (x: W untransformed, r: R1, z: NZ) of the [NZ][R1][S0] view of one transform, S0 = nx * R0 (everything faster
// than the digit r); tile coordinate c0 = group of W adjacent elements of S0 (W * sizeof = one 128-byte segment), o = batch item
template <typename T, int S0, int R1, int NZ, int W, int NT_, bool HALF_, int OCC_, typename RLY, typename RLZ, bool SPLIT_OUT_ = false>
struct PairYZ {
    static constexpr int P = W * R1 * NZ, NT = NT_, OCC = OCC_, WIDTH = W;
    static constexpr bool HALF = HALF_, TWOUT = false, SPLIT_IN = false, SPLIT_OUT = SPLIT_OUT_;
    using MAP = PairMap<W, R1, NZ, /*GI*/ 1, S0, (long long)S0 * R1, /*GO*/ 1, S0, (long long)S0 * R1,
                        /*C*/ S0 / W, 1, /*BI*/ W, 0, /*BO*/ W, 0, /*outer: one transform*/ (long long)S0 * R1 * NZ>;
    using SY = typename Nd2AxisStages<1, R1, W, 1, RLY, Nd2StageList<>>::type;
    using SZ = typename Nd2AxisStages<2, NZ, W * R1, 1, RLZ, Nd2StageList<>>::type;
    using SL = typename Nd2Concat<SY, SZ>::type;
};

template <typename T, typename CFG> static inline int launch_pair(const PairArgs* a, hipStream_t s) {
    if (a->tiles <= 0) return 0;
    if (a->tiles > 2147483647ll) return -1;
    hipLaunchKernelGGL((fft_pair_kernel<T, CFG>), dim3((unsigned)a->tiles), dim3(CFG::NT), 0, s, *a);
    return (int)hipGetLastError();
}

}  // namespace mifft

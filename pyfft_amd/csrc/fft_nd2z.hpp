// Whole 2-D / 3-D transforms of a fixed shape whose points fit no two-work-groups-per-CU tile (32768 points fp32 / 16384 fp64: the
// "huge" tiles of fft_nd2.hpp hold one transform per CU -- 512 threads x 64 points, nothing overlaps a tile's load, stages and store:
// 0.46-0.54 of the roofline at 1 GiB, and at the reference's 32 MiB protocol (16, 16, 128) x 128 occupies half of the CUs: 0.315).
//
// Round 5: TWO work-groups per transform, no exchange between them.  One decimation-in-frequency step along the SLOWEST axis s of
// length LS = 2 H (pyfft/kernel.mako:805-1047 is the same algebra with R = 2, M = H):
//     X[2 k + p] = FFT_H( u_p ),   u_0[j] = x[j] + x[j + H],   u_1[j] = (x[j] - x[j + H]) * w(LS)^j,      j < H along s
// Work-group (t, p) reads ALL of transform t -- both halves, combined into u_p as the operands arrive -- runs the remaining
// (H x rest)-point transform on the "big" tile form (half the points: two work-groups per CU, half-exchange stages, 32 / 16 points
// per thread) and stores the planes (rows) k_s = 2 k + p of the result: whole contiguous planes, every output byte written once.
// The input is read twice; the pair (t, 0), (t, 1) is given block indices 8 apart, so it lands on the same XCD back to back and the
// second reader finds the lines in that XCD's L2 (work-groups go to the XCDs round robin: tools/xcd_probe.hip) -- HBM sees them once.
// Out of place only (a work-group overwrites planes its partner has not read yet); the launcher keeps in-place calls on the huge tile.
//
// RS work-groups per transform (RS = 4): the same step with radix RS -- u_p[j] = (sum_r x[j + r Q] w(RS)^(r p)) w(LS)^(p j), Q = LS / RS,
// X[RS k + p] = FFT_Q(u_p) -- for shapes of RS two-per-CU tiles (65536 points fp32 / 32768 fp64: (256, 256), fp64 (16, 16, 128) ...),
// which otherwise take TWO launches (one crossing of the L2 <-> fabric path instead of two; the input is read RS times out of the L2).
#pragma once
#include "fft_nd2.hpp"

namespace mifft {

// LX, LY, LZ: the FULL shape (x contiguous); the split axis is z for 3-D shapes, y for 2-D ones.
template <typename T, int LX, int LY, int LZ, int NT_, bool HALF_, int OCC_, typename RLX, typename RLY, typename RLZ, int RS_ = 2>
struct Nd2zCfg {
    static constexpr bool SPLIT_Z = LZ > 1;
    static constexpr int RS = RS_;                                          // work-groups per transform
    static constexpr int LS = SPLIT_Z ? LZ : LY, H = LS / RS;
    static constexpr int HY = SPLIT_Z ? LY : H, HZ = SPLIT_Z ? H : 1;      // the half transform's shape (LX, HY, HZ)
    static constexpr int P = LX * HY * HZ, NT = NT_, OCC = OCC_;
    static constexpr int SLAB = P / H;                                      // points per index of the split axis
    static constexpr bool HALF = HALF_;
    static_assert(LS >= 2 * RS && LX > 1 && (P % NT) == 0 && (RS == 2 || RS == 4 || RS == 8), "bad shape");
    using SX = typename Nd2AxisStages<0, LX, 1, 1, RLX, Nd2StageList<>>::type;
    using SY = typename Nd2AxisStages<1, HY, LX, 1, RLY, Nd2StageList<>, (SPLIT_Z ? 1 : RS)>::type;
    using SZ = typename Nd2AxisStages<2, HZ, LX * HY, 1, RLZ, Nd2StageList<>, RS>::type;
    using SL = typename Nd2Concat<typename Nd2Concat<SX, SY>::type, SZ>::type;
};

template <typename T, typename CFG>
__global__ void __launch_bounds__(CFG::NT) __attribute__((amdgpu_waves_per_eu(CFG::OCC))) fft_nd2z_kernel(const TileArgs a) {
    constexpr int P = CFG::P, NT = CFG::NT, PPT = P / NT, SLAB = CFG::SLAB, RS = CFG::RS;
    constexpr bool HALF = CFG::HALF;
    using SL = typename CFG::SL;
    using First = Nd2Stage<T, P, NT, HALF, typename Nd2First<SL>::type>;
    static_assert(First::AX == 0 && First::SA == 1, "the first stage runs along x: the operands of a butterfly share their split-axis index");
    using LdsT = typename std::conditional<HALF, T, cplx<T>>::type;
    __shared__ __attribute__((aligned(16))) LdsT lds[P + P / 16];
    const int tid = threadIdx.x;
    // blocks b, b + 8, ... = the RS parts of one transform: same XCD, dispatched back to back
    const unsigned b = blockIdx.x;
    const long long t = (long long)(b / (8u * RS)) * 8 + (b & 7u);
    const unsigned p = (b >> 3) % (unsigned)RS;
    if (t * ((long long)RS * P) >= a.total) return;
    const char* inb = reinterpret_cast<const char*>(reinterpret_cast<const cplx<T>*>(a.in0) + t * ((long long)RS * P));
    char* outb = reinterpret_cast<char*>(reinterpret_cast<cplx<T>*>(a.out0) + t * ((long long)RS * P) + (long long)p * SLAB);
    const cplx<T>* tw[3] = {reinterpret_cast<const cplx<T>*>(a.tw_L), reinterpret_cast<const cplx<T>*>(a.tw_lo),
                            reinterpret_cast<const cplx<T>*>(a.tw_hi)};
    const cplx<T>* tws = CFG::SPLIT_Z ? tw[2] : tw[1];          // w(LS)^k, the split axis' own table
    const T sx = (T)a.scale;
    const T sy = a.inverse ? -sx : sx;
    const T sgn = p ? (T)-1 : (T)1;
    const T cj = a.inverse ? (T)-1 : (T)1;
    cplx<T> v[PPT];
    // the radix-RS step's own factors w(RS)^(r p), r >= 1 (block-uniform; from the split axis' table: w(RS)^m = w(LS)^(m LS / RS))
    cplx<T> cr[RS];
    if constexpr (RS > 2) {
        static_for<RS - 1>([&](auto rr) {
            constexpr int r = rr + 1;
            cr[r] = tws[((r * p) % (unsigned)RS) * (unsigned)(CFG::LS / RS)];
        });
    }
    // first-stage operands: u_p = (sum_r x[e + r P] w(RS)^(r p)) * w(LS)^(p * js); the inverse transform conjugates its input first
    static_for<First::NB>([&](auto bb) {
        constexpr int bq = bb;
        int base, jb;
        First::geom(bq, tid, base, jb);
        const unsigned voff = (unsigned)(base + jb) * (unsigned)sizeof(cplx<T>);
        cplx<T> w = {(T)1, (T)0};
        if (p) w = tws[p * (unsigned)(base / SLAB)];
        static_for<First::R>([&](auto kk) {
            constexpr int k = kk;
            const char* q = inb + (size_t)(k * First::LR) * sizeof(cplx<T>) + voff;
            const cplx<T> x0 = *reinterpret_cast<const cplx<T>*>(q);
            cplx<T> u;
            if constexpr (RS == 2) {
                const cplx<T> x1 = *reinterpret_cast<const cplx<T>*>(q + (size_t)P * sizeof(cplx<T>));
                u.x = x0.x + sgn * x1.x;
                u.y = (x0.y + sgn * x1.y) * cj;
            } else {
                u.x = x0.x;
                u.y = x0.y * cj;
                static_for<RS - 1>([&](auto rr) {
                    constexpr int r = rr + 1;
                    cplx<T> xr = *reinterpret_cast<const cplx<T>*>(q + (size_t)r * P * sizeof(cplx<T>));
                    xr.y *= cj;
                    u.x += xr.x * cr[r].x - xr.y * cr[r].y;
                    u.y += xr.x * cr[r].y + xr.y * cr[r].x;
                });
            }
            v[bq * First::R + k] = cmul<T>(u, w);
        });
    });
    const int nt_out = (a.nt & 4) ? 2 : ((a.nt & 2) ? 1 : 0);
    // last stage (along the split axis, SA = SLAB): result index k of the part's transform is plane RS k + p of the whole one
    auto sink = [&](auto st, cplx<T>* r) __attribute__((always_inline)) {
        using St = decltype(st);
        static_assert(St::SA == SLAB && St::LA == CFG::H, "the last stage runs along the split axis");
        auto stores = [&](auto ntc) __attribute__((always_inline)) {
            constexpr int NTS = ntc;
            static_for<St::NB>([&](auto bb) {
                constexpr int bq = bb;
                int base, jb;
                St::geom(bq, tid, base, jb);
                const unsigned voff = (unsigned)(base + St::idxd(jb) * RS * SLAB) * (unsigned)sizeof(cplx<T>);
                static_for<St::R>([&](auto kk) {
                    constexpr int k = kk;
                    cplx<T> o = r[bq * St::R + k];
                    o.x *= sx;
                    o.y *= sy;
                    char* kb = outb + (size_t)(k * St::Ns) * (RS * SLAB) * sizeof(cplx<T>);
                    cplx<T>* q = reinterpret_cast<cplx<T>*>(kb + voff);
                    if constexpr (NTS == 2) store_wt<T>(kb, voff, o);
                    else if constexpr (NTS == 1) __builtin_nontemporal_store(o, q);
                    else *q = o;
                });
            });
        };
        if (nt_out == 2) stores(IC<2>{});
        else if (nt_out == 1) stores(IC<1>{});
        else stores(IC<0>{});
    };
    nd2_chain_sink<T, P, NT, HALF, true, SL>(lds, v, tw, tid, sink);
}

template <typename T, typename CFG> static inline int launch_nd2z(const TileArgs* a, hipStream_t s) {
    const long long ntrans = a->total / ((long long)CFG::RS * CFG::P);
    if (ntrans <= 0) return 0;
    const long long blocks = ((ntrans + 7) / 8) * 8 * CFG::RS;
    if (blocks > 2147483647ll) return -1;
    hipLaunchKernelGGL((fft_nd2z_kernel<T, CFG>), dim3((unsigned)blocks), dim3(CFG::NT), 0, s, *a);
    return (int)hipGetLastError();
}

}  // namespace mifft

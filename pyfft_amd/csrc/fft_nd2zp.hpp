// Several work-groups per transform (fft_nd2z.hpp) on DENSE SPLIT-COMPLEX batches: re / im planes on both sides, 16 bytes per lane and
// plane on either side of HBM (round 6).
//
// The one-tile-per-CU shapes of the reference's float32 dtype (pyfft/plan.py:26-35: two scalar planes) -- (16, 16, 128): 32768 points,
// 256 KiB of registers + LDS, one work-group per CU -- ran 0.45 of the roofline on fft_nd2p.hpp where their interleaved twins run 0.64 on
// two half-size work-groups per transform (profiles/r06_f_planes_probe.log).  Here the same split: work-group (t, p) reads ALL of transform
// t, folds the radix-RS decimation-in-frequency step along the slowest axis into the operands as they arrive
//     u_p[j] = (sum_r x[j + r Q] w(RS)^(r p)) w(LS)^(p j),   Q = LS / RS,      X[RS k + p] = FFT_Q(u_p)
// (pyfft/kernel.mako:805-1047 with R = RS, M = Q), transforms its part on the two-per-CU tile form and stores the planes k_s = RS k + p.
//   in    16 bytes per lane of the re plane and of the im plane for each of the RS parts (x-adjacent points), combined in registers,
//         u_p -> LDS at its natural position in the part -> the first stage fetches its operands (the linear-load form of fft_nd2p.hpp)
//   out   the last stage spills to natural positions; every lane reads VEC x-adjacent results back and stores 16 bytes per plane
// Same stage lists, tables and butterflies as the interleaved kernel of the shape (Nd2zCfg), the decimation step in the same operation
// order: the same values to rounding.  Out of place only, like fft_nd2z.hpp (a work-group overwrites planes its partners still read).
#pragma once
#include "fft_nd2z.hpp"

namespace mifft {

template <typename T, typename CFG>
__global__ void __launch_bounds__(CFG::NT) __attribute__((amdgpu_waves_per_eu(CFG::OCC))) fft_nd2zp_kernel(const TileArgs a) {
    constexpr int P = CFG::P, NT = CFG::NT, PPT = P / NT, SLAB = CFG::SLAB, RS = CFG::RS;
    constexpr bool HALF = CFG::HALF;
    constexpr int VEC = 16 / (int)sizeof(T);
    constexpr int NV = PPT / VEC;
    static_assert(PPT % VEC == 0 && SLAB % VEC == 0 && (RS == 2 || RS == 4), "bad tile");
    using SL = typename CFG::SL;
    using First = Nd2Stage<T, P, NT, HALF, typename Nd2First<SL>::type>;
    using LdsT = typename std::conditional<HALF, T, cplx<T>>::type;
    using VT = T __attribute__((ext_vector_type(VEC)));
    __shared__ __attribute__((aligned(16))) LdsT lds[P + P / 16];
    int tid = threadIdx.x;
    // blocks b, b + 8, ... = the RS parts of one transform: same XCD, dispatched back to back (fft_nd2z.hpp)
    const unsigned b = blockIdx.x;
    const long long t = (long long)(b / (8u * RS)) * 8 + (b & 7u);
    const unsigned p = (b >> 3) % (unsigned)RS;
    if (t * ((long long)RS * P) >= a.total) return;
    const T* in_re = reinterpret_cast<const T*>(a.in0) + t * ((long long)RS * P);
    const T* in_im = reinterpret_cast<const T*>(a.in1) + t * ((long long)RS * P);
    T* out_re = reinterpret_cast<T*>(a.out0) + t * ((long long)RS * P) + (long long)p * SLAB;
    T* out_im = reinterpret_cast<T*>(a.out1) + t * ((long long)RS * P) + (long long)p * SLAB;
    const cplx<T>* tw[3] = {reinterpret_cast<const cplx<T>*>(a.tw_L), reinterpret_cast<const cplx<T>*>(a.tw_lo),
                            reinterpret_cast<const cplx<T>*>(a.tw_hi)};
    const cplx<T>* tws = CFG::SPLIT_Z ? tw[2] : tw[1];          // w(LS)^k, the split axis' own table
    const T sx = (T)a.scale;
    const T sy = a.inverse ? -sx : sx;
    const T sgn = p ? (T)-1 : (T)1;
    const T cj = a.inverse ? (T)-1 : (T)1;
    const bool nt_in = (a.nt & 1) != 0;
    cplx<T> cr[RS];
    if constexpr (RS > 2) {
        static_for<RS - 1>([&](auto rr) {
            constexpr int r = rr + 1;
            cr[r] = tws[((r * p) % (unsigned)RS) * (unsigned)(CFG::LS / RS)];
        });
    }
    cplx<T> v[PPT];

    // ---- planes -> u_p in registers -> LDS (natural positions of the part) -> the first stage's operands
    {
        VT re[NV], im[NV];
        auto ld = [&](const T* q) __attribute__((always_inline)) -> VT {
            return nt_in ? __builtin_nontemporal_load(reinterpret_cast<const VT*>(q)) : *reinterpret_cast<const VT*>(q);
        };
        static_for<NV>([&](auto ii) {
            constexpr int it = ii;
            const unsigned e = (unsigned)(it * NT + tid) * (unsigned)VEC;       // VEC x-adjacent points of one index js = e / SLAB of the split axis
            cplx<T> w = {(T)1, (T)0};
            if (p) w = tws[p * (e / (unsigned)SLAB)];
            VT ur, ui;
            if constexpr (RS == 2) {
                const VT r0 = ld(in_re + e), i0 = ld(in_im + e), r1 = ld(in_re + e + P), i1 = ld(in_im + e + P);
                ur = r0 + sgn * r1;
                ui = (i0 + sgn * i1) * cj;
            } else {
                ur = ld(in_re + e);
                ui = ld(in_im + e) * cj;
                static_for<RS - 1>([&](auto rr) {
                    constexpr int r = rr + 1;
                    const VT xr = ld(in_re + e + r * P), xi = ld(in_im + e + r * P) * cj;
                    ur += xr * cr[r].x - xi * cr[r].y;
                    ui += xr * cr[r].y + xi * cr[r].x;
                });
            }
            // u * w, component by component in the operation order of cmul (fft_butterfly.hpp)
            static_for<VEC>([&](auto jj) {
                constexpr int j = jj;
                cplx<T> u;
                u.x = ur[j];
                u.y = ui[j];
                u = cmul<T>(u, w);
                re[it][j] = u.x;
                im[it][j] = u.y;
            });
        });
        if constexpr (!HALF) {
            static_for<NV>([&](auto ii) {
                constexpr int it = ii;
                LdsT* q = lds + row2_pad((it * NT + tid) * VEC);
                static_for<VEC>([&](auto jj) {
                    constexpr int j = jj;
                    cplx<T> c;
                    c.x = re[it][j];
                    c.y = im[it][j];
                    q[j] = c;
                });
            });
            __syncthreads();
            First::template fetch<0>(lds, v, tid);
        } else {
            static_for<NV>([&](auto ii) {
                constexpr int it = ii;
                LdsT* q = lds + row2_pad((it * NT + tid) * VEC);
                static_for<VEC>([&](auto jj) {
                    constexpr int j = jj;
                    q[j] = re[it][j];
                });
            });
            __syncthreads();
            First::template fetch<1>(lds, v, tid);
            __syncthreads();
            static_for<NV>([&](auto ii) {
                constexpr int it = ii;
                LdsT* q = lds + row2_pad((it * NT + tid) * VEC);
                static_for<VEC>([&](auto jj) {
                    constexpr int j = jj;
                    q[j] = im[it][j];
                });
            });
            __syncthreads();
            First::template fetch<2>(lds, v, tid);
        }
        __syncthreads();
    }

    // ---- the stages; the last one's results go back through LDS to their natural positions in the part: index k of the split axis is
    //      plane RS k + p of the whole transform
    const int nts = (a.nt & 4) ? 2 : ((a.nt & 2) ? 1 : 0);       // stores: 2 write-through (small launches), 1 non-temporal, 0 plain
    auto put = [&](T* plane, unsigned o, VT w) __attribute__((always_inline)) {
        if (nts == 2) store_vec_wt(reinterpret_cast<VT*>(plane + o), w);
        else if (nts == 1) __builtin_nontemporal_store(w, reinterpret_cast<VT*>(plane + o));
        else *reinterpret_cast<VT*>(plane + o) = w;
    };
    auto sink = [&](auto stc, const cplx<T>* vv) __attribute__((always_inline)) {
        using St = decltype(stc);
        static_assert(St::SA == SLAB && St::LA == CFG::H, "the last stage runs along the split axis");
        int t2 = tid;
        asm volatile("" : "+v"(t2));
        __syncthreads();              // everybody has fetched its operands of this stage
        if constexpr (!HALF) {
            St::template spill<0>(lds, vv, t2);
            __syncthreads();
            static_for<NV>([&](auto ii) {
                constexpr int it = ii;
                const unsigned e = (unsigned)(it * NT + t2) * (unsigned)VEC;
                const unsigned o = (e / (unsigned)SLAB) * (unsigned)(RS * SLAB) + (e % (unsigned)SLAB);
                const LdsT* q = lds + row2_pad((it * NT + t2) * VEC);
                VT r, m;
                static_for<VEC>([&](auto jj) {
                    constexpr int j = jj;
                    const cplx<T> c = q[j];
                    r[j] = c.x * sx;
                    m[j] = c.y * sy;
                });
                put(out_re, o, r);
                put(out_im, o, m);
            });
        } else {
            static_for<2>([&](auto cc) {
                constexpr int comp = cc;
                if constexpr (comp == 1) __syncthreads();      // the real parts have been read
                St::template spill<comp + 1>(lds, vv, t2);
                __syncthreads();
                static_for<NV>([&](auto ii) {
                    constexpr int it = ii;
                    const unsigned e = (unsigned)(it * NT + t2) * (unsigned)VEC;
                    const unsigned o = (e / (unsigned)SLAB) * (unsigned)(RS * SLAB) + (e % (unsigned)SLAB);
                    const LdsT* q = lds + row2_pad((it * NT + t2) * VEC);
                    VT w;
                    static_for<VEC>([&](auto jj) {
                        constexpr int j = jj;
                        w[j] = q[j] * (comp == 0 ? sx : sy);
                    });
                    put(comp == 0 ? out_re : out_im, o, w);
                });
            });
        }
    };
    asm volatile("" : "+v"(tid));
    nd2_chain_sink<T, P, NT, HALF, true, SL>(lds, v, tw, tid, sink);
}

template <typename T, typename CFG> static inline int launch_nd2zp(const TileArgs* a, hipStream_t s) {
    const long long ntrans = a->total / ((long long)CFG::RS * CFG::P);
    if (ntrans <= 0) return 0;
    const long long blocks = ((ntrans + 7) / 8) * 8 * CFG::RS;
    if (blocks > 2147483647ll) return -1;
    hipLaunchKernelGGL((fft_nd2zp_kernel<T, CFG>), dim3((unsigned)blocks), dim3(CFG::NT), 0, s, *a);
    return (int)hipGetLastError();
}

}  // namespace mifft

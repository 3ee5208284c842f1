// Whole 2-D / 3-D transforms of one fixed shape on DENSE SPLIT-COMPLEX batches (re / im planes on both sides): the stage chain of
// fft_nd2.hpp with 16 bytes per lane and plane on either side of HBM (round 6).
//
// The reference's float32 / float64 dtypes are first-class (pyfft/plan.py:26-35: two scalar planes instead of interleaved pairs;
// BASELINE.json names the layout).  Rounds 4-5 sent a dense split-complex N-D plan through the TILED kernel of fft_nd2t.hpp with one
// tile per "parent": its first stage loads and its last stage stores one SCALAR per lane and plane (4 bytes in fp32: 256 bytes per wave
// instruction where the interleaved kernel moves 512), and every access goes through the tiling's run-time address arithmetic.  The
// published shapes ran 0.69-0.86 of their interleaved twins at 1 GiB per side and 0.44-0.77 at the reference's 32 MiB
// (profiles/r04_bb_reference_shapes_split_final.log).
//
// Here a lane moves VEC = 16 / sizeof(T) x-adjacent scalars of one plane per instruction, on both sides:
//   in    VEC reals + VEC imaginaries of the x-adjacent points 4q .. 4q+3 (fp64: 2q, 2q+1) of the work-group's dense tile -> LDS at their
//         natural positions -> the first stage fetches its operands (the linear-load form of fft_nd2.hpp: one more exchange than the
//         register-edged form, which interleaved shapes with short x rows take as well)
//   out   the last stage spills its results to their natural positions like any other stage; every lane reads VEC x-adjacent results back
//         and stores 16 bytes to the re plane and 16 to the im plane (HALF tiles: the real parts, then the imaginary parts -- the
//         exchange they run between any two stages)
// Same butterflies, twiddles and stage lists as the interleaved kernel of the shape (Nd2Auto), so the same values to rounding.
#pragma once
#include "fft_nd2.hpp"

namespace mifft {

template <typename T, int LX, int LY, int LZ, int P, int NT, bool HALF, int OCC, typename RLX, typename RLY, typename RLZ>
__global__ void __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(OCC))) fft_nd2p_kernel(const TileArgs a) {
    constexpr int PPT = P / NT;
    constexpr int VEC = 16 / (int)sizeof(T);
    constexpr int NV = PPT / VEC;
    static_assert(PPT * NT == P && P % (LX * LY * LZ) == 0 && PPT % VEC == 0 && LX % VEC == 0, "bad tile");
    using SX = typename Nd2AxisStages<0, LX, 1, 1, RLX, Nd2StageList<>>::type;
    using SY = typename Nd2AxisStages<1, LY, LX, 1, RLY, Nd2StageList<>>::type;
    using SZ = typename Nd2AxisStages<2, LZ, LX * LY, 1, RLZ, Nd2StageList<>>::type;
    using SL = typename Nd2Concat<typename Nd2Concat<SX, SY>::type, SZ>::type;
    using First = Nd2Stage<T, P, NT, HALF, typename Nd2First<SL>::type>;
    using LdsT = typename std::conditional<HALF, T, cplx<T>>::type;
    using VT = T __attribute__((ext_vector_type(VEC)));
    __shared__ __attribute__((aligned(16))) LdsT lds[P + P / 16];
    int tid = threadIdx.x;
    const long long g0 = (long long)blockIdx.x * P;
    const long long left = a.total - g0;       // points from the start of the tile to the end of the data (whole transforms)
    const T* in_re = reinterpret_cast<const T*>(a.in0) + g0;
    const T* in_im = reinterpret_cast<const T*>(a.in1) + g0;
    T* out_re = reinterpret_cast<T*>(a.out0) + g0;
    T* out_im = reinterpret_cast<T*>(a.out1) + g0;
    const cplx<T>* tw[3] = {reinterpret_cast<const cplx<T>*>(a.tw_L), reinterpret_cast<const cplx<T>*>(a.tw_lo),
                            reinterpret_cast<const cplx<T>*>(a.tw_hi)};
    const T sx = (T)a.scale;
    const T sy = a.inverse ? -sx : sx;
    const T csign = a.inverse ? (T)-1 : (T)1;
    const bool nt_in = (a.nt & 1) != 0;
    cplx<T> v[PPT];

    // ---- planes -> LDS (natural positions) -> the first stage's operands
    {
        VT re[NV], im[NV];
        static_for<NV>([&](auto ii) {
            constexpr int it = ii;
            const unsigned e = (unsigned)(it * NT + tid) * (unsigned)VEC;
            VT r = {}, m = {};
            if ((long long)e < left) {
                if (nt_in) {
                    r = __builtin_nontemporal_load(reinterpret_cast<const VT*>(in_re + e));
                    m = __builtin_nontemporal_load(reinterpret_cast<const VT*>(in_im + e));
                } else {
                    r = *reinterpret_cast<const VT*>(in_re + e);
                    m = *reinterpret_cast<const VT*>(in_im + e);
                }
            }
            re[it] = r;
            im[it] = m * csign;
        });
        if constexpr (!HALF) {
            static_for<NV>([&](auto ii) {
                constexpr int it = ii;
                LdsT* p = lds + row2_pad((it * NT + tid) * VEC);     // (VEC consecutive points never straddle a padding slot: VEC divides 16)
                static_for<VEC>([&](auto jj) {
                    constexpr int j = jj;
                    cplx<T> c;
                    c.x = re[it][j];
                    c.y = im[it][j];
                    p[j] = c;
                });
            });
            __syncthreads();
            First::template fetch<0>(lds, v, tid);
        } else {
            static_for<NV>([&](auto ii) {
                constexpr int it = ii;
                LdsT* p = lds + row2_pad((it * NT + tid) * VEC);
                static_for<VEC>([&](auto jj) {
                    constexpr int j = jj;
                    p[j] = re[it][j];
                });
            });
            __syncthreads();
            First::template fetch<1>(lds, v, tid);
            __syncthreads();
            static_for<NV>([&](auto ii) {
                constexpr int it = ii;
                LdsT* p = lds + row2_pad((it * NT + tid) * VEC);
                static_for<VEC>([&](auto jj) {
                    constexpr int j = jj;
                    p[j] = im[it][j];
                });
            });
            __syncthreads();
            First::template fetch<2>(lds, v, tid);
        }
        __syncthreads();
    }

    // ---- the stages; the last one's results go back through LDS to their natural positions, then out in 16-byte runs per plane
    const int nts = (a.nt & 4) ? 2 : ((a.nt & 2) ? 1 : 0);       // stores: 2 write-through (small launches), 1 non-temporal, 0 plain
    auto put = [&](T* plane, unsigned e, VT w) __attribute__((always_inline)) {
        if (nts == 2) store_vec_wt(reinterpret_cast<VT*>(plane + e), w);
        else if (nts == 1) __builtin_nontemporal_store(w, reinterpret_cast<VT*>(plane + e));
        else *reinterpret_cast<VT*>(plane + e) = w;
    };
    auto sink = [&](auto stc, const cplx<T>* vv) __attribute__((always_inline)) {
        using St = decltype(stc);
        int t2 = tid;
        asm volatile("" : "+v"(t2));
        __syncthreads();              // everybody has fetched its operands of this stage
        if constexpr (!HALF) {
            St::template spill<0>(lds, vv, t2);
            __syncthreads();
            static_for<NV>([&](auto ii) {
                constexpr int it = ii;
                const unsigned e = (unsigned)(it * NT + t2) * (unsigned)VEC;
                const LdsT* p = lds + row2_pad((it * NT + t2) * VEC);
                VT r, m;
                static_for<VEC>([&](auto jj) {
                    constexpr int j = jj;
                    const cplx<T> c = p[j];
                    r[j] = c.x * sx;
                    m[j] = c.y * sy;
                });
                if ((long long)e < left) {
                    put(out_re, e, r);
                    put(out_im, e, m);
                }
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
                    const LdsT* p = lds + row2_pad((it * NT + t2) * VEC);
                    VT w;
                    static_for<VEC>([&](auto jj) {
                        constexpr int j = jj;
                        w[j] = p[j] * (comp == 0 ? sx : sy);
                    });
                    if ((long long)e < left) put(comp == 0 ? out_re : out_im, e, w);
                });
            });
        }
    };
    asm volatile("" : "+v"(tid));
    nd2_chain_sink<T, P, NT, HALF, true, SL>(lds, v, tw, tid, sink);
}

// the tile configuration Nd2Auto derives for the interleaved kernel of the same shape
template <typename T, int X, int Y, int Z> static inline int launch_nd2p_auto(const TileArgs* a, hipStream_t s) {
    using C = Nd2Auto<T, X, Y, Z>;
    const long long tiles = (a->total + C::P - 1) / C::P;
    if (tiles <= 0) return 0;
    if (tiles > 2147483647ll) return -1;
    hipLaunchKernelGGL((fft_nd2p_kernel<T, X, Y, Z, C::P, C::NT, C::HALF, C::OCC, typename C::RLX, typename C::RLY, typename C::RLZ>),
                       dim3((unsigned)tiles), dim3(C::NT), 0, s, *a);
    return (int)hipGetLastError();
}

}  // namespace mifft

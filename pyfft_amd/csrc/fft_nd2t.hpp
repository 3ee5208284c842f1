// Whole 2-D / 3-D transforms of one fixed shape on TILES OF A BIGGER ARRAY ("2D/3D tiled batch support ... to transform several
// tiles of big 2D/3D array in one pass", the reference's TODO.txt:6-7): the fixed-shape stage chain of fft_nd2.hpp with the
// transforms of a work-group taken from / written to their places in the parent array, so that the gather and the scatter
// (pyfft_amd/generic.py: two streaming copies around a dense N-D plan, three HBM round trips) cost nothing: ONE round trip.
//
// A work-group's P points are P / (LX*LY*LZ) whole tiles; tile number g = item * (cx*cy*cz) + (iz*cy + iy)*cx + ix sits at
//     item * parent + iz*LZ*pitch_z + iy*LY*pitch_y + ix*LX        (elements; x contiguous, pitch_y = parent x extent, ...)
// and its point (z, y, x) at + z*pitch_z + y*pitch_y + x.  Every stage works inside one tile, so the k-th operand of a
// butterfly is a fixed multiple of (1 | pitch_y | pitch_z) away from the first.  In place or out of place (same geometry on
// both sides); interleaved data, or split planes (SPLIT: in0 / in1 = re / im planes of the parent, same element offsets -- round 4).
#pragma once
#include "fft_nd2.hpp"

namespace mifft {

struct TiledGeom {
    long long pitch_y, pitch_z;   // elements between consecutive y / z of the PARENT array
    long long parent;             // elements per parent array (batch item)
    long long tiles;              // total number of tiles (transforms)
    int cx, cy, cz;               // tiles per parent axis
};

template <int LX, int LY, int LZ> struct Nd2tAddr {
    static constexpr int N = LX * LY * LZ;
    // element offset of tile-local point e of the work-group whose first tile is g0; returns -1 beyond the last tile
    static __device__ __forceinline__ long long at(const TiledGeom& g, long long g0, int e) {
        const long long t = g0 + e / N;
        if (t >= g.tiles) return -1;
        const int r = e % N, x = r % LX, y = (r / LX) % LY, z = r / (LX * LY);
        const int per = g.cx * g.cy * g.cz;
        const long long item = t / per;
        const int rem = (int)(t % per), ix = rem % g.cx, iy = (rem / g.cx) % g.cy, iz = rem / (g.cx * g.cy);
        return item * g.parent + ((long long)iz * LZ + z) * g.pitch_z + ((long long)iy * LY + y) * g.pitch_y + (long long)ix * LX + x;
    }
};

// SPLIT: planes on the input side; SPLIT_OUT: on the output side (default: both alike; planes in / interleaved out = the plane pass of a
// split-complex multi-pass plan, whose temp buffer is interleaved)
template <typename T, int LX, int LY, int LZ, int P, int NT, bool HALF, int OCC, bool EDGE_IN, typename RLX, typename RLY, typename RLZ, bool SPLIT = false,
          bool SPLIT_OUT = SPLIT>
__global__ void __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(OCC))) fft_nd2t_kernel(const TileArgs a, const TiledGeom g) {
    constexpr int PPT = P / NT;
    static_assert(PPT * NT == P && P % (LX * LY * LZ) == 0 && PPT % 2 == 0, "bad tile");
    static_assert(EDGE_IN || !HALF, "the linear-load form is built for full-complex exchanges only");
    using SX = typename Nd2AxisStages<0, LX, 1, 1, RLX, Nd2StageList<>>::type;
    using SY = typename Nd2AxisStages<1, LY, LX, 1, RLY, Nd2StageList<>>::type;
    using SZ = typename Nd2AxisStages<2, LZ, LX * LY, 1, RLZ, Nd2StageList<>>::type;
    using SL = typename Nd2Concat<typename Nd2Concat<SX, SY>::type, SZ>::type;
    using First = Nd2Stage<T, P, NT, HALF, typename Nd2First<SL>::type>;
    using LdsT = typename std::conditional<HALF, T, cplx<T>>::type;
    using Addr = Nd2tAddr<LX, LY, LZ>;
    __shared__ __attribute__((aligned(16))) LdsT lds[P + P / 16];
    const int tid = threadIdx.x;
    const long long g0 = (long long)blockIdx.x * (P / Addr::N);
    const cplx<T>* in = reinterpret_cast<const cplx<T>*>(a.in0);
    cplx<T>* out = reinterpret_cast<cplx<T>*>(a.out0);
    const T* in_re = reinterpret_cast<const T*>(a.in0);
    const T* in_im = reinterpret_cast<const T*>(a.in1);
    T* out_re = reinterpret_cast<T*>(a.out0);
    T* out_im = reinterpret_cast<T*>(a.out1);
    const cplx<T>* tw[3] = {reinterpret_cast<const cplx<T>*>(a.tw_L), reinterpret_cast<const cplx<T>*>(a.tw_lo),
                            reinterpret_cast<const cplx<T>*>(a.tw_hi)};
    const long long pitch[3] = {1, g.pitch_y, g.pitch_z};
    const T sx = (T)a.scale;
    const T sy = a.inverse ? -sx : sx;
    cplx<T> v[PPT];
    if constexpr (EDGE_IN) {
        static_for<First::NB>([&](auto bb) {
            constexpr int b = bb;
            int base, jb;
            First::geom(b, tid, base, jb);
            const long long off = Addr::at(g, g0, base + jb * First::SA);
            const long long step = (long long)First::LR * pitch[First::AX];
            static_for<First::R>([&](auto kk) {
                constexpr int k = kk;
                cplx<T> p;
                p.x = 0; p.y = 0;
                if (off >= 0) {
                    if constexpr (SPLIT) {
                        p.x = in_re[off + k * step];
                        p.y = in_im[off + k * step];
                    } else {
                        p = in[off + k * step];
                    }
                }
                v[b * First::R + k] = p;
            });
        });
    } else {
        // 16-byte linear load (thread: tile-local points 2*(it*NT + tid), +1: the same row of the same tile), then one exchange
        using V4 = T __attribute__((ext_vector_type(4)));
        static_for<PPT / 2>([&](auto ii) {
            constexpr int it = ii;
            const long long off = Addr::at(g, g0, (it * NT + tid) * 2);
            V4 q = {0, 0, 0, 0};
            if constexpr (SPLIT) {
                // two x-adjacent points of each plane (tile rows start on even elements: LX and the pitches are even)
                using V2 = T __attribute__((ext_vector_type(2)));
                if (off >= 0) {
                    const V2 re = *reinterpret_cast<const V2*>(in_re + off), im = *reinterpret_cast<const V2*>(in_im + off);
                    q.x = re.x; q.y = im.x; q.z = re.y; q.w = im.y;
                }
            } else {
                if (off >= 0) q = *reinterpret_cast<const V4*>(in + off);
            }
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
    {
        const T csign = a.inverse ? (T)-1 : (T)1;
        static_for<PPT>([&](auto i) { v[i].y *= csign; });
    }
    // small launches (MIFFT_FLAG_WRITE_THROUGH -> nt bit 2, round 5): agent-scope write-through stores, so that the result does not wait
    // dirty in the L2s for the end-of-kernel write-back (planes at the reference's 32 MiB: fp64 (16, 16) 0.50 -> 0.79 of the roofline,
    // profiles/r05_fp64_write_through_rows.log); a scalar of a plane goes out as a relaxed agent-scope atomic store (= global_store sc1)
    const bool wt = (a.nt & 4) != 0;
    auto sink = [&](auto stc, const cplx<T>* vv) __attribute__((always_inline)) {
        using St = decltype(stc);
        using Bits = typename std::conditional<sizeof(T) == 4, unsigned, unsigned long long>::type;
        auto stores = [&](auto wtc) __attribute__((always_inline)) {
            constexpr bool WT = decltype(wtc)::value != 0;
            static_for<St::NB>([&](auto bb) {
                constexpr int b = bb;
                int base, jb;
                St::geom(b, tid, base, jb);
                const long long off = Addr::at(g, g0, base + St::idxd(jb) * St::SA);
                const long long step = (long long)St::Ns * pitch[St::AX];
                if (off >= 0) {
                    static_for<St::R>([&](auto kk) {
                        constexpr int k = kk;
                        cplx<T> p = vv[b * St::R + k];
                        p.x *= sx;
                        p.y *= sy;
                        if constexpr (SPLIT_OUT) {
                            if constexpr (WT) {
                                // (scalars first: __builtin_bit_cast of the vector ELEMENT p.y took element 0 -- both planes got the real part)
                                const T pre = p.x, pim = p.y;
                                __hip_atomic_store(reinterpret_cast<Bits*>(out_re + off + k * step), __builtin_bit_cast(Bits, pre), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                __hip_atomic_store(reinterpret_cast<Bits*>(out_im + off + k * step), __builtin_bit_cast(Bits, pim), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            } else {
                                out_re[off + k * step] = p.x;
                                out_im[off + k * step] = p.y;
                            }
                        } else {
                            if constexpr (WT) store_wt_ptr<T>(out + off + k * step, p);
                            else out[off + k * step] = p;
                        }
                    });
                }
            });
        };
        if (wt) stores(IC<1>{});
        else stores(IC<0>{});
    };
    nd2_chain_sink<T, P, NT, HALF, true, SL>(lds, v, tw, tid, sink);
}

// launch with the tile configuration Nd2Auto derives for the dense kernel of the same shape
template <typename T, int X, int Y, int Z, bool SPLIT = false, bool SPLIT_OUT = SPLIT>
static inline int launch_nd2t_auto(const TileArgs* a, const TiledGeom* g, hipStream_t s) {
    using C = Nd2Auto<T, X, Y, Z>;
    const long long per_wg = C::P / (X * Y * Z);
    const long long wgs = (g->tiles + per_wg - 1) / per_wg;
    if (wgs <= 0) return 0;
    if (wgs > 2147483647ll) return -1;
    hipLaunchKernelGGL((fft_nd2t_kernel<T, X, Y, Z, C::P, C::NT, C::HALF, C::OCC, C::EDGE_IN, typename C::RLX, typename C::RLY, typename C::RLZ, SPLIT, SPLIT_OUT>),
                       dim3((unsigned)wgs), dim3(C::NT), 0, s, *a, *g);
    return (int)hipGetLastError();
}

}  // namespace mifft

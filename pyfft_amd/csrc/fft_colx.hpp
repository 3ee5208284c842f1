// Strided-axis (COL) pass on the register-edged stage chain of fft_nd2.hpp, for radices whose tile fits no other COL kernel:
// fp64 L = 2048 (8 columns x 2048 rows = 256 KiB of points: 1024 threads x 16 points, exchanges in half form through 136 KiB
// of LDS, one work-group per CU).  With it a long fp64 axis of 2^21 / 2^22 points takes two passes instead of three
// (pyfft/kernel.py:259-283 factors a long axis the same way, in base 128).
//
// Pass algebra (SURVEY.md 3.3 / pyfft/kernel.mako:805-1047), one tile = W adjacent columns of the [L][M*S] matrix:
//     out[l][q][j] = scale * w(L*M)^(l*q) * sum_r in[r][l][j] * w(L)^(r*q)
// Tile-local space [L][W] (column fastest): the first radix stage takes its operands straight from HBM (the lanes run along
// the W columns first: 128-byte row segments), the last one stores straight from registers -- for the transposing form
// (S == 1) a wave covers 8 columns x 8 consecutive q, i.e. 128-byte runs of the output rows.
// Forms: TR (S == 1, M >= W) and plain (S >= W, any M); the tile is W whole columns of ONE matrix, interleaved on both
// sides.  Everything else (tiny S, split planes, ragged column counts) takes the generic tile kernel.
#pragma once
#include "fft_nd2.hpp"

namespace mifft {

// nt: bit 0 non-temporal loads, bit 1 non-temporal stores (run-time: only the load block and the store block exist twice -- two
// whole copies of the tile code behind one branch made the compiler carry both copies' live ranges, 60 spilled registers);
// WT: write-through stores (the intermediate of a fused two-pass kernel).  `lds` holds P + P / 16 scalars (HALF) or complex numbers.
template <typename T, int L, int W, int NT, bool HALF, bool TR, bool TW, bool WT, typename RL, typename LdsT>
__device__ __forceinline__ void colx_tile(const TileArgs& a, const long long o_in, const long long o_out, const long long rem0,
                                          LdsT* lds, const int nt) {
    constexpr int P = L * W, PPT = P / NT;
    using SL = typename Nd2AxisStages<1, L, W, 1, RL, Nd2StageList<>>::type;
    using First = Nd2Stage<T, P, NT, HALF, typename Nd2First<SL>::type>;
    static_assert(First::AX == 1 && First::SA == W && First::LA == L, "stage list of the strided axis");
    // (opaque copies: inside the persistent loop of fft_fusedx_f64.hip the wave-uniform offsets derived from the shifts and the
    // per-thread ones derived from the thread index are recomputed per tile instead of being hoisted out of the loop for both
    // tile kinds and spilled -- as in fft_col2.hpp)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    int logMS = a.logMS, logS = a.logS;
    asm volatile("" : "+s"(logMS), "+s"(logS));
    const cplx<T>* tw[3] = {nullptr, reinterpret_cast<const cplx<T>*>(a.tw_L), nullptr};

    // ---- first stage: v[b*R + k] = in[(jb + k*LR) rows][column c]
    cplx<T> v[PPT];
    {
        const char* src = reinterpret_cast<const char*>(reinterpret_cast<const cplx<T>*>(a.in0) + (o_in * a.ostride_in + rem0));
        auto loads = [&](auto ntc) __attribute__((always_inline)) {
            constexpr bool NTIN = decltype(ntc)::value != 0;
            static_for<First::NB>([&](auto bb) {
                constexpr int b = bb;
                int base, jb;
                First::geom(b, tid, base, jb);      // base = column c (the tile is one [L][W] block), jb = row of the first operand
                const unsigned voff = (((unsigned)jb << logMS) + (unsigned)base) * (unsigned)sizeof(cplx<T>);
                static_for<First::R>([&](auto kk) {
                    constexpr int k = kk;
                    const char* p = src + (((long long)(k * First::LR)) << logMS) * (long long)sizeof(cplx<T>);
                    if constexpr (NTIN) v[b * First::R + k] = __builtin_nontemporal_load(reinterpret_cast<const cplx<T>*>(p + voff));
                    else v[b * First::R + k] = *reinterpret_cast<const cplx<T>*>(p + voff);
                });
            });
        };
        if (nt & 1) loads(IC<1>{}); else loads(IC<0>{});
    }
    {   // (a multiplication, not a branch: the two-sided copy of the imaginary parts a branch needs costs 32 registers here)
        const T csign = a.inverse ? (T)-1 : (T)1;
        static_for<PPT>([&](auto i) { v[i].y *= csign; });
    }

    // ---- the stages; the last one stores through `sink`
    const T sx = (T)a.scale;
    const T sy = a.inverse ? -sx : sx;
    const long long l0 = rem0 >> logS;
    const long long jp0 = rem0 & ((1ll << logS) - 1);
    const cplx<T>* twlo = reinterpret_cast<const cplx<T>*>(a.tw_lo);
    const cplx<T>* twhi = reinterpret_cast<const cplx<T>*>(a.tw_hi);
    const int tw_shift = a.tw_shift;
    const unsigned lomask = (1u << tw_shift) - 1u;
    char* const dst = reinterpret_cast<char*>(reinterpret_cast<cplx<T>*>(a.out0) +
                                              (TR ? (a.ostride_out * o_out + rem0 * L) : (a.ostride_out * o_out + ((l0 * L) << logS) + jp0)));
    auto stores = [&](auto stc, auto policy, const cplx<T>* vv) __attribute__((always_inline)) {
        using St = decltype(stc);
        constexpr int ST = decltype(policy)::value;   // 0 plain, 1 non-temporal, 2 write-through
        static_for<St::NB>([&](auto bb) {
            constexpr int b = bb;
            int base, jb;
            St::geom(b, tid, base, jb);          // base = column c, results q = idxd(jb) + k * Ns
            const unsigned q0 = (unsigned)St::idxd(jb);
            const unsigned c = (unsigned)base;
            // TR: out[(rem0 + c) * L + q]; plain: out[((l0 * L + q) << logS) + jp0 + c]
            const unsigned voff = (TR ? (c * (unsigned)L + q0) : ((q0 << logS) + c)) * (unsigned)sizeof(cplx<T>);
            const unsigned l = TR ? (unsigned)rem0 + c : (unsigned)l0;
            // inter-pass twiddle w(L * M)^(l * q), q = q0 + k * Ns (round 4, as fft_pair.hpp): every fourth factor from the two-level
            // table, the three behind it one multiplication each by the step w^(l * Ns) (rounds 2-3: every factor looked up)
            cplx<T> wstep = {(T)1, (T)0}, wcur = {(T)1, (T)0};
            if constexpr (TW && St::R > 1) {
                const unsigned es = l * (unsigned)St::Ns;
                wstep = cmul<T>(twlo[es & lomask], twhi[es >> tw_shift]);
            }
            static_for<St::R>([&](auto kk) {
                constexpr int k = kk;
                cplx<T> p = vv[b * St::R + k];
                if constexpr (TW) {
                    if constexpr (k % 4 == 0) {
                        const unsigned e = l * (q0 + (unsigned)(k * St::Ns));
                        wcur = cmul<T>(twlo[e & lomask], twhi[e >> tw_shift]);
                    } else {
                        wcur = cmul<T>(wcur, wstep);
                    }
                    p = cmul<T>(p, wcur);
                }
                p.x *= sx;
                p.y *= sy;
                char* kb = dst + (TR ? (long long)(k * St::Ns) : ((long long)(k * St::Ns) << logS)) * (long long)sizeof(cplx<T>);
                if constexpr (ST == 2) store_wt<T>(kb, voff, p);
                else if constexpr (ST == 1) __builtin_nontemporal_store(p, reinterpret_cast<cplx<T>*>(kb + voff));
                else *reinterpret_cast<cplx<T>*>(kb + voff) = p;
            });
        });
    };
    auto sink = [&](auto stc, const cplx<T>* vv) __attribute__((always_inline)) {
        if constexpr (WT) stores(stc, IC<2>{}, vv);
        else if (nt & 4) stores(stc, IC<2>{}, vv);
        else if (nt & 2) stores(stc, IC<1>{}, vv);
        else stores(stc, IC<0>{}, vv);
    };
    nd2_chain_sink<T, P, NT, HALF, true, SL>(lds, v, tw, tid, sink);
}

template <typename T, int L, int W, int NT, bool HALF, int OCC, bool TR, bool TW, typename RL>
__global__ void __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(OCC))) fft_colx_kernel(const TileArgs a) {
    using LdsT = typename std::conditional<HALF, T, cplx<T>>::type;
    __shared__ __attribute__((aligned(16))) LdsT lds[L * W + L * W / 16];
    const long long col0 = (long long)blockIdx.x * W;
    const long long o = col0 >> a.logMS;
    const long long rem0 = col0 & ((1ll << a.logMS) - 1);
    colx_tile<T, L, W, NT, HALF, TR, TW, false, RL>(a, o, o, rem0, lds, a.nt);
}

}  // namespace mifft

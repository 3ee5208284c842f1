// Persistent 2-D kernel for SPLIT-COMPLEX fp32 planes, ROW FIRST (second batch of round 4).
//
// The 2-D form of fft_fused2.hpp runs a 2-D transform as two TRANSPOSING column passes; its first pass reads the user's planes in
// 16-column tiles -- 64 bytes per row and plane, half of a 128-byte line, and every line crosses the fabric twice (the split note in
// fft_fused2.hpp).  For split planes the chain's own order is the better one: pass 0 transforms the x axis of WHOLE contiguous rows
// (4 KiB per plane and row at nx = 1024: full lines; the register-edged row stages of fft_row2.hpp with their first-stage operands
// loaded from the two planes) and writes ring[y][kx] interleaved, write-through; pass 1 transforms the y axis of 16 adjacent kx
// (fft_col2.hpp: 128-byte interleaved segments out of the last-level cache) and writes out[ky][kx] to the planes -- 64-byte segments,
// but on the WRITE side, where a partial line costs no second crossing.
//   pass-0 item (t, tile): rows y = 16 * tile ... + 15 of transform t, W of them at a time (the row kernel's own W rows per work-group)
//   pass-1 item (t, tile): columns kx = 16 * tile ... + 15, all ky
//   dependencies as in fft_fused2.hpp: a pass-1 tile reads every row of its transform (wdone[t] == ny / 16), a pass-0 tile waits until
//   the ring slot's previous owner has been read (rdone).
#pragma once
#include "fft_fused2.hpp"
#include "fft_row2.hpp"

namespace mifft {

// row configuration of the x axis (the plain launch's: fft_row_f32.hip): rows per group, threads per row = 256 / W, radix list
template <int NX> struct Fused2rRow;
template <> struct Fused2rRow<256> { static constexpr int W = 8; using RL = RadixList<8, 8, 4>; };
template <> struct Fused2rRow<512> { static constexpr int W = 8; using RL = RadixList<16, 2, 16>; };
template <> struct Fused2rRow<1024> { static constexpr int W = 4; using RL = RadixList<16, 4, 16>; };
template <typename RL> struct Fused2rFirst;
template <int R, int... Rest> struct Fused2rFirst<RadixList<R, Rest...>> { static constexpr int value = R; };

// AY = ny / 256, NX = nx: fp32, split-complex user planes, interleaved ring
template <int NX, int AY>
__global__ void __launch_bounds__(256, 2) fft_fused2r_kernel(const FusedArgs f) {
    using T = float;
    using Row = Fused2rRow<NX>;
    constexpr int NY = 256 * AY, W = Row::W, TPR = 256 / W, PPT = NX / TPR, LP = NX + NX / 16;
    constexpr int R = Fused2rFirst<typename Row::RL>::value, LR = NX / R;
    using Stages = Row2Stages<T, NX, TPR, 1, true, false, typename Row::RL>;
    constexpr int E0 = W * LP, E1 = Col2Lds<AY, false, sizeof(cplx<T>)>::ELEMS;
    __shared__ __attribute__((aligned(16))) cplx<T> lds[E0 > E1 ? E0 : E1];
    __shared__ unsigned s_item;
    constexpr unsigned t0 = NY / 16, t1 = NX / 16;
    constexpr unsigned per0 = t0 >= t1 ? t0 / t1 : 1u, per1 = t1 > t0 ? t1 / t0 : 1u;
    fused_loop<per0, per1, false>(
        f.c, &s_item,
        [&](unsigned t, unsigned slot, unsigned tile, auto) {
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));
            const int c = tid / TPR, u = tid % TPR;
            const T* re = reinterpret_cast<const T*>(f.p0.in0);
            const T* im = reinterpret_cast<const T*>(f.p0.in1);
            cplx<T>* ring = reinterpret_cast<cplx<T>*>(f.p0.out0);
#pragma nounroll
            for (int g = 0; g < 16 / W; ++g) {
                if (g) __syncthreads();            // the previous group's last-stage operands have left the LDS
                const long long y = (long long)tile * 16 + g * W + c;
                const long long rin = ((long long)t * NY + y) * NX, rout = ((long long)slot * NY + y) * NX;
                cplx<T> v[PPT];
                // first-stage operands (Row2Stages::load_regs order): v[b * R + k] = row[b * TPR + k * LR + u], one scalar per plane
                static_for<PPT>([&](auto ii) {
                    constexpr int i = ii, b = i / R, k = i % R;
                    v[i].x = __builtin_nontemporal_load(re + rin + b * TPR + k * LR + u);   // streamed once: the cache belongs to the ring
                    v[i].y = __builtin_nontemporal_load(im + rin + b * TPR + k * LR + u);
                });
                char* outb = reinterpret_cast<char*>(ring + rout + u);
                Stages::template run<true>(lds + c * LP, v, f.p0, u, nullptr, outb, 0u, true);
            }
        },
        [&](unsigned slot, unsigned t, unsigned tile, auto hook) {
            col2_tile<T, AY, false, false, false, false, false, true, true>(f.p1, (long long)slot, (long long)t, (long long)tile * 16, lds, hook);
        });
}

}  // namespace mifft

// Contiguous-axis (ROW) transforms, register-edged form: one work-group owns W rows of L points (whole LDS-resident
// transform; counterpart of pyfft/kernel.mako:725-803), TPR = NT / W threads per row, every thread keeps PPT = L / TPR
// points of its row.
// Unlike fft_tile.hpp the first radix stage takes its operands straight from HBM into registers and the last one
// stores its results straight from registers to HBM, so a three-stage transform crosses LDS twice instead of four times
// (the long rows are LDS-bound: one 70 / 139 KiB tile per CU).
//
// Stockham autosort indexing as in fft_tile.hpp: stage with radix R and Ns = product of the earlier radices reads
// idx = j + k*L/R, multiplies by w(L)^(k * (j mod Ns) * L/(Ns*R)) and writes (j & ~(Ns-1))*R + (j & (Ns-1)) + k*Ns.
// For the first stage (Ns = 1) the reads are x[j + k*L/R]: consecutive j across the lanes, so coalesced; for the last
// stage (Ns = L/R) the writes are out[j + k*L/R], coalesced likewise.
#pragma once
#include <type_traits>
#include "fft_tile.hpp"

namespace mifft {

// v[k] *= w(L)^(k*ai), k < R, with few look-ups (the table competes with the data stream for the L1) and few live
// registers: powers 1..7 (or 1..R-1 for R <= 8) by a product tree from one look-up, every further block of 8 from its own
// look-up w(L)^(8h*ai) times the first seven -- no power is further than 4 products from a table entry.
// TS: the table holds w(TS * L)^k -- every TS-th entry is used (round 5: the half-length stages of fft_nd2z.hpp read the full axis' table)
template <typename T, int R, int TS = 1> __device__ __forceinline__ void row2_twiddle(const cplx<T>* twL, int ai, cplx<T>* v) {
    constexpr int Q = R < 8 ? R : 8;
    cplx<T> t[Q];
    t[1] = twL[ai * TS];
    static_for<Q - 2>([&](auto kk) {
        constexpr int k = kk + 2;
        if constexpr ((k & 1) == 0) t[k] = cmul<T>(t[k / 2], t[k / 2]);
        else t[k] = cmul<T>(t[k - 1], t[1]);
    });
    static_for<Q - 1>([&](auto kk) {
        constexpr int k = kk + 1;
        v[k] = cmul<T>(v[k], t[k]);
    });
    static_for<R / 8 - (R >= 8 ? 1 : 0)>([&](auto hh) {
        constexpr int h = (hh + 1) * 8;
        const cplx<T> th = twL[h * ai * TS];
        v[h] = cmul<T>(v[h], th);
        static_for<7>([&](auto kk) {
            constexpr int k = kk + 1;
            v[h + k] = cmul<T>(v[h + k], cmul<T>(th, t[k]));
        });
    });
}

__device__ __forceinline__ int row2_pad(int i) { return i + (i >> 4); }

// TPR = threads per row; `tid` below is the thread's index within its row, `lds` the row's own LDS slab.
// HALF: the exchange between two stages moves the real parts, then the imaginary parts, through an LDS slab of L scalars
// instead of L complex numbers: twice the barriers, half the LDS, so twice the work-groups per CU for the longest rows.
// LAY (second batch of round 4): bit 0 = the input is two scalar planes (inb / inb1 = the row's first real / imaginary scalar), bit 1 =
// the output likewise (outb / outb1); 0 = interleaved complex numbers on both sides.
template <typename T, int L, int TPR, int Ns, bool FIRST, bool HALF, typename RL, int LAY = 0> struct Row2Stages;

template <typename T, int L, int TPR, int Ns, bool FIRST, bool HALF, int R, int... Rest, int LAY>
struct Row2Stages<T, L, TPR, Ns, FIRST, HALF, RadixList<R, Rest...>, LAY> {
    static constexpr int NT = TPR;
    static constexpr int PPT = L / NT;
    static constexpr int NB = PPT / R;
    static constexpr int LR = L / R;
    static constexpr bool LAST = sizeof...(Rest) == 0;
    static_assert(NB >= 1 && NB * R == PPT, "radix must divide the points per thread");
    static_assert(!(FIRST && LAST), "needs at least two stages");
    using LdsT = typename std::conditional<HALF, T, cplx<T>>::type;

    // operands of this stage: LDS -> v (component sel: 0 = whole complex number, 1 = real part, 2 = imaginary part)
    // LDS addresses are one per-thread base + a compile-time offset (DS instructions carry a 16-bit immediate): with
    // idx = base + c, c a multiple of a power of two that divides 16 or is divided by it and base % 16 + c % 16 < 16,
    // pad(idx) = pad(base) + c + (c >> 4).  Computing pad() per access instead keeps one address VGPR per point alive.
    template <int SEL> static __device__ __forceinline__ void fetch(const LdsT* lds, cplx<T>* v, int tid) {
        static_for<NB>([&](auto bb) {
            constexpr int b = bb;
            const LdsT* p = lds + row2_pad(b * NT + tid);  // j = b * NT + tid < LR
            static_for<R>([&](auto kk) {
                constexpr int k = kk;
                constexpr int off = k * LR + ((k * LR) >> 4);
                if constexpr (SEL == 0) v[b * R + k] = p[off];
                else if constexpr (SEL == 1) v[b * R + k].x = p[off];
                else v[b * R + k].y = p[off];
            });
        });
    }
    // results of this stage: v -> LDS at the autosort position
    template <int SEL> static __device__ __forceinline__ void spill(LdsT* lds, const cplx<T>* v, int tid) {
        static_for<NB>([&](auto bb) {
            constexpr int b = bb;
            const int j = b * NT + tid;
            const int idxD = (j & ~(Ns - 1)) * R + (j & (Ns - 1));
            LdsT* p = lds + row2_pad(idxD);
            static_for<R>([&](auto kk) {
                constexpr int k = kk;
                constexpr int off = k * Ns + ((k * Ns) >> 4);
                if constexpr (SEL == 0) p[off] = v[b * R + k];
                else if constexpr (SEL == 1) p[off] = v[b * R + k].x;
                else p[off] = v[b * R + k].y;
            });
        });
    }

    // v holds the operands of this stage (from HBM for the first stage, from the exchange otherwise)
    // Global addresses are (base pointer) + constant + (32-bit per-thread byte offset voff): with one row per
    // work-group the base is wave-uniform and lives in SGPRs, so a load costs no address VGPRs beyond voff (with 64-bit
    // per-load addresses the 32 loads of a thread alone pinned 64 VGPRs and halved the occupancy).
    // operands of the FIRST stage for registers [I0, I0 + CNT): v[b*R + k] = in[b*NT + k*LR (+ tid)], straight from HBM
    template <int I0, int CNT, bool NTL>
    static __device__ __forceinline__ void load_regs(cplx<T>* v, const char* inb, unsigned voff) {
        static_assert(FIRST, "first-stage addressing");
        static_for<CNT>([&](auto ii) {
            constexpr int i = I0 + ii, b = i / R, k = i % R;
            const cplx<T>* p = reinterpret_cast<const cplx<T>*>(inb + (size_t)(b * NT + k * LR) * sizeof(cplx<T>) + voff);
            if constexpr (NTL) v[i] = __builtin_nontemporal_load(p);
            else v[i] = *p;
        });
    }

    // the same operands from two scalar planes (voff = the thread's byte offset within a plane)
    template <int I0, int CNT, bool NTL>
    static __device__ __forceinline__ void load_regs_split(cplx<T>* v, const char* inre, const char* inim, unsigned voff) {
        static_assert(FIRST, "first-stage addressing");
        static_for<CNT>([&](auto ii) {
            constexpr int i = I0 + ii, b = i / R, k = i % R;
            const T* pr = reinterpret_cast<const T*>(inre + (size_t)(b * NT + k * LR) * sizeof(T) + voff);
            const T* pi = reinterpret_cast<const T*>(inim + (size_t)(b * NT + k * LR) * sizeof(T) + voff);
            if constexpr (NTL) {
                v[i].x = __builtin_nontemporal_load(pr);
                v[i].y = __builtin_nontemporal_load(pi);
            } else {
                v[i].x = *pr;
                v[i].y = *pi;
            }
        });
    }

    // PRELOADED: v already holds the row (persistent form: loaded behind the previous row's stores).
    // FirstStage / (next_inb, next_valid): persistent form only -- the LAST stage issues the NEXT row's first-stage loads into
    // the registers of every butterfly right behind that butterfly's stores, so that they fly under the remaining
    // butterflies and stores of this row (one row fills the CU: there is no second work-group to overlap with).
    template <bool PRELOADED = false, typename FirstStage = void>
    static __device__ __forceinline__ void run(LdsT* lds, cplx<T>* v, const TileArgs& a, int tid, const char* inb,
                                               char* outb, unsigned voff, bool valid, const char* next_inb = nullptr,
                                               bool next_valid = false, const char* inb1 = nullptr, char* outb1 = nullptr) {
        const cplx<T>* twL = reinterpret_cast<const cplx<T>*>(a.tw_L);
        if constexpr (FIRST) {
            if constexpr (!PRELOADED) {
                // all loads of the row in flight before the first butterfly
                static_for<PPT>([&](auto i) { v[i].x = 0; v[i].y = 0; });
                if (valid) {
                    // MIFFT_FLAG_STREAM_SRC: the input is read once (first pass of a multi-pass plan)
                    if constexpr (LAY & 1) {     // planes: voff counts complex numbers' bytes, a plane's scalar is half as wide
                        if (a.nt & 1) load_regs_split<0, PPT, true>(v, inb, inb1, voff / 2);
                        else load_regs_split<0, PPT, false>(v, inb, inb1, voff / 2);
                    } else {
                        if (a.nt & 1) load_regs<0, PPT, true>(v, inb, voff);
                        else load_regs<0, PPT, false>(v, inb, voff);
                    }
                }
            }
            if (a.inverse) static_for<PPT>([&](auto i) { v[i].y = -v[i].y; });
        }
        if constexpr (LAST && !std::is_void<FirstStage>::value) {
            // persistent form (plain accesses): butterfly by butterfly -- twiddle, radix-R, stores, then the next row's
            // first-stage loads into the same registers
            const T sx = (T)a.scale;
            const T sy = a.inverse ? -sx : sx;
            static_for<NB>([&](auto bb) {
                constexpr int b = bb;
                const int j = b * NT + tid;
                const int ai = (j & (Ns - 1)) * (L / (Ns * R));
                row2_twiddle<T, R>(twL, ai, v + b * R);
                Dft<R, T>::run(v + b * R);
                static_for<R>([&](auto kk) {
                    constexpr int k = kk;
                    cplx<T> p = v[b * R + k];
                    p.x *= sx;
                    p.y *= sy;
                    *reinterpret_cast<cplx<T>*>(outb + (size_t)(b * NT + k * Ns) * sizeof(cplx<T>) + voff) = p;
                });
                if (next_valid) FirstStage::template load_regs<b * R, R, false>(v, next_inb, voff);
                else static_for<R>([&](auto kk) { v[b * R + kk] = cplx<T>{(T)0, (T)0}; });   // ends the live range
            });
            return;
        }
        static_for<NB>([&](auto bb) {
            constexpr int b = bb;
            const int j = b * NT + tid;
            if constexpr (!FIRST) {
                const int ai = (j & (Ns - 1)) * (L / (Ns * R));
                row2_twiddle<T, R>(twL, ai, v + b * R);
            }
            Dft<R, T>::run(v + b * R);
        });
        if constexpr (LAST) {
            const T sx = (T)a.scale;
            const T sy = a.inverse ? -sx : sx;
            if (valid) {
                auto stores = [&](auto ntc) __attribute__((always_inline)) {
                    constexpr int NTS = ntc;   // 0 plain, 1 non-temporal, 2 write-through
                    static_for<NB>([&](auto bb) {
                        constexpr int b = bb;  // Ns == LR: idxD = j = b * NT + tid
                        static_for<R>([&](auto kk) {
                            constexpr int k = kk;
                            cplx<T> p = v[b * R + k];
                            p.x *= sx;
                            p.y *= sy;
                            if constexpr (LAY & 2) {
                                T* qr = reinterpret_cast<T*>(outb + (size_t)(b * NT + k * Ns) * sizeof(T) + voff / 2);
                                T* qi = reinterpret_cast<T*>(outb1 + (size_t)(b * NT + k * Ns) * sizeof(T) + voff / 2);
                                if constexpr (NTS == 2) {          // write-through: agent-scope stores of the two scalars
                                    __hip_atomic_store(qr, p.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    __hip_atomic_store(qi, p.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                } else if constexpr (NTS == 1) {
                                    __builtin_nontemporal_store(p.x, qr);
                                    __builtin_nontemporal_store(p.y, qi);
                                } else {
                                    *qr = p.x;
                                    *qi = p.y;
                                }
                            } else {
                            char* kb = outb + (size_t)(b * NT + k * Ns) * sizeof(cplx<T>);
                            cplx<T>* q = reinterpret_cast<cplx<T>*>(kb + voff);
                            if constexpr (NTS == 2) store_wt_ptr<T>(kb + voff, p);   // (the row base may differ across the wave)
                            else if constexpr (NTS == 1) __builtin_nontemporal_store(p, q);
                            else *q = p;
                            }
                        });
                    });
                };
                if (a.nt & 4) stores(IC<2>{}); else if (a.nt & 2) stores(IC<1>{}); else stores(IC<0>{});
            }
        } else {
            using Next = Row2Stages<T, L, TPR, Ns * R, false, HALF, RadixList<Rest...>, LAY>;
            if constexpr (!FIRST) __syncthreads();  // everybody has fetched its operands of this stage
            if constexpr (!HALF) {
                spill<0>(lds, v, tid);
                __syncthreads();
                Next::template fetch<0>(lds, v, tid);
            } else {
                // every step gets its own opaque copy of the thread index: otherwise the LDS base addresses of a
                // layout are computed once and kept in VGPRs across the barriers and the other component's step
                auto fresh = [&]() __attribute__((always_inline)) {
                    int t = tid;
                    asm volatile("" : "+v"(t));
                    return t;
                };
                spill<1>(lds, v, fresh());
                __syncthreads();
                Next::template fetch<1>(lds, v, fresh());  // the real-part registers are free again: reuse them
                __syncthreads();
                spill<2>(lds, v, fresh());
                __syncthreads();
                Next::template fetch<2>(lds, v, fresh());
            }
            Next::template run<PRELOADED, FirstStage>(lds, v, a, tid, inb, outb, voff, valid, next_inb, next_valid, inb1, outb1);
        }
    }
};

// OCC: waves per SIMD the register allocation must leave room for (1 = whatever the kernel needs)
template <typename T, int L, int W, int NT, bool HALF, int OCC, typename RL, int LAY = 0>
__global__ void __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(OCC))) fft_row2_kernel(const TileArgs a) {
    constexpr int TPR = NT / W;
    constexpr int PPT = L / TPR;
    constexpr int LP = L + L / 16;
    static_assert(TPR * W == NT && PPT * TPR == L && L >= 16, "bad row configuration");
    using LdsT = typename std::conditional<HALF, T, cplx<T>>::type;
    __shared__ __attribute__((aligned(16))) LdsT lds[W * LP];
    const int c = W == 1 ? 0 : threadIdx.x / TPR, u = W == 1 ? threadIdx.x : threadIdx.x % TPR;
    const long long row = (long long)blockIdx.x * W + c;
    const bool valid = row < a.total;
    // (planes: the row's first real / imaginary scalar; the thread's offset below counts complex numbers' bytes and is halved at the access)
    const char* inb = (LAY & 1) ? reinterpret_cast<const char*>(reinterpret_cast<const T*>(a.in0) + row * a.ostride_in)
                                : reinterpret_cast<const char*>(reinterpret_cast<const cplx<T>*>(a.in0) + row * a.ostride_in);
    const char* inb1 = (LAY & 1) ? reinterpret_cast<const char*>(reinterpret_cast<const T*>(a.in1) + row * a.ostride_in) : nullptr;
    char* outb = (LAY & 2) ? reinterpret_cast<char*>(reinterpret_cast<T*>(a.out0) + row * a.ostride_out)
                           : reinterpret_cast<char*>(reinterpret_cast<cplx<T>*>(a.out0) + row * a.ostride_out);
    char* outb1 = (LAY & 2) ? reinterpret_cast<char*>(reinterpret_cast<T*>(a.out1) + row * a.ostride_out) : nullptr;
    unsigned voff = (unsigned)u * (unsigned)sizeof(cplx<T>);
    if constexpr (W > 1) {  // the row differs across the wave: fold the thread's offset into its own 64-bit base
        inb += (LAY & 1) ? voff / 2 : voff;
        if constexpr (LAY & 1) inb1 += voff / 2;
        outb += (LAY & 2) ? voff / 2 : voff;
        if constexpr (LAY & 2) outb1 += voff / 2;
        voff = 0;
    }
    cplx<T> v[PPT];
    Row2Stages<T, L, TPR, 1, true, HALF, RL, LAY>::run(lds + c * LP, v, a, u, inb, outb, voff, valid, nullptr, false, inb1, outb1);
}

// Persistent form for the rows that fill a CU (W == 1, one work-group per CU or two): the work-group walks rows
// blockIdx.x, blockIdx.x + gridDim.x, ...; the first-stage loads of the next row are issued from the last stage of the
// current one (Row2Stages::run, FirstStage).
template <typename T, int L, int NT, bool HALF, int OCC, typename RL>
__global__ void __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(OCC))) fft_row2p_kernel(const TileArgs a) {
    constexpr int PPT = L / NT;
    constexpr int LP = L + L / 16;
    using LdsT = typename std::conditional<HALF, T, cplx<T>>::type;
    using First = Row2Stages<T, L, NT, 1, true, HALF, RL>;
    __shared__ __attribute__((aligned(16))) LdsT lds[LP];
    const unsigned voff = (unsigned)threadIdx.x * (unsigned)sizeof(cplx<T>);
    long long row = blockIdx.x;
    if (row >= a.total) return;
    cplx<T> v[PPT];
    {
        const char* inb = reinterpret_cast<const char*>(reinterpret_cast<const cplx<T>*>(a.in0) + row * a.ostride_in);
        First::template load_regs<0, PPT, false>(v, inb, voff);
    }
    for (;;) {
        const long long next = row + gridDim.x;
        // (the strides are laundered per row so that the ~PPT row offsets are not hoisted out of the loop as SGPR pairs)
        long long sin = a.ostride_in, sout = a.ostride_out;
        asm volatile("" : "+s"(sin), "+s"(sout));
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const char* inb = reinterpret_cast<const char*>(reinterpret_cast<const cplx<T>*>(a.in0) + row * sin);
        char* outb = reinterpret_cast<char*>(reinterpret_cast<cplx<T>*>(a.out0) + row * sout);
        const char* nin = reinterpret_cast<const char*>(reinterpret_cast<const cplx<T>*>(a.in0) + next * sin);
        First::template run<true, First>(lds, v, a, tid, inb, outb, (unsigned)tid * (unsigned)sizeof(cplx<T>), true, nin, next < a.total);
        if (next >= a.total) break;
        row = next;
        __syncthreads();   // the last exchange's LDS reads are over before the next row's first spill
    }
}

// blocks_per_cu: resident work-groups per CU of this configuration (LDS- or register-bound); the grid is that many per CU
template <typename T, int L, int NT, typename RL, bool HALF, int OCC>
static inline int launch_row2p(const TileArgs* a, hipStream_t s, int blocks_per_cu) {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    }
    long long grid = (long long)cus * blocks_per_cu;
    if (grid > a->total) grid = a->total;
    if (grid <= 0) return 0;
    hipLaunchKernelGGL((fft_row2p_kernel<T, L, NT, HALF, OCC, RL>), dim3((unsigned)grid), dim3(NT), 0, s, *a);
    return (int)hipGetLastError();
}

template <typename T, int L, int W, int NT, typename RL, bool HALF = false, int OCC = 1, int LAY = 0>
static inline int launch_row2(const TileArgs* a, hipStream_t s, int query_only) {
    if (query_only) return 0;
    const long long tiles = (a->total + W - 1) / W;
    if (tiles <= 0) return 0;
    if (tiles > 2147483647ll) return -1;
    hipLaunchKernelGGL((fft_row2_kernel<T, L, W, NT, HALF, OCC, RL, LAY>), dim3((unsigned)tiles), dim3(NT), 0, s, *a);
    return (int)hipGetLastError();
}

// the same configuration for the layout of the pass at hand: interleaved, planes -> planes (a split-complex single-pass plan), planes
// -> interleaved (the first pass of a split-complex multi-pass plan, whose temp buffer is interleaved)
template <typename T, int L, int W, int NT, typename RL, bool HALF = false, int OCC = 1>
static inline int launch_row2_lay(const TileArgs* a, hipStream_t s, int query_only) {
    if (query_only || (!a->split && !a->split_out)) return launch_row2<T, L, W, NT, RL, HALF, OCC, 0>(a, s, query_only);
    if (a->split && a->split_out) return launch_row2<T, L, W, NT, RL, HALF, OCC, 3>(a, s, 0);
    if (a->split) return launch_row2<T, L, W, NT, RL, HALF, OCC, 1>(a, s, 0);
    return -2;      // (interleaved -> planes: no plan has a contiguous-axis pass last but one; the LDS-staged kernel takes it)
}

}  // namespace mifft

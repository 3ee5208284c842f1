// XCD-cooperative single-crossing kernel for a long contiguous fp32 axis N = 1024 * 1024 (BASELINE config 2).
//
// Why: every two-pass form whose inter-pass intermediate leaves the XCD moves 4 units over the L2 <-> fabric
// links for 2 algorithmic units (rocprofv3: 2.006 x, profiles/r01_l_c2_pmc_summary.json) and is capped at ~41 %
// of the 8 TB/s roofline (fft_fused2.hpp reaches 37.3 %).  Here the 64 work-groups that the dispatcher places on
// one XCD (2 per CU x 32 CUs; found with HW_REG_XCC_ID, never assumed) own ONE transform at a time:
//     pass 0   work-group r loads columns [16r, 16r+16) of the [1024][1024] input view straight into registers
//              (64 points per thread, the col2 data flow of fft_col2.hpp), radix 16 * 4 * 16 with one LDS exchange
//     hand-off the 8 MiB intermediate never leaves the chiplet: it is exchanged all-to-all among the 64
//              work-groups in FOUR rounds of 2 MiB through that XCD's own L2 (plain stores -> vmcnt drain ->
//              per-producer flag; the consumer polls its 16 producers' flags, then sc1 = L1-bypassing loads).
//              In round k work-group r sends the radix-16 slab qa = (r - k) mod 4 -- 16 values per thread, one
//              to each consumer r' = qb0*4 + qa -- and receives the 16 inputs of ITS first-stage butterfly
//              a = (r + k) mod 4 of pass 1 into the registers the slab just vacated (producer and consumer
//              thread indices coincide), so a transform lives in the XCD's VGPR file between its two HBM
//              crossings.  r mod 4 is a template parameter (four code paths) so every register index is static.
//     pass 1   radix 16 * 4 * 16 on the received columns, stored straight from registers; the loads of the XCD's
//              next transform are issued slab by slab into the registers the stores vacate.
// HBM traffic is the algorithmic 16 B per point; XCDs run unsynchronised, so one XCD's exchange/butterfly phases
// overlap the other XCDs' HBM phases (one XCD alone streams at 1.3 TB/s, 1.75 x its share of the chip's copy rate:
// tools/xcd_probe.hip, profiles/r02_a_xcd_probe.log).
//
// Arithmetic is the col2 arithmetic operation for operation, so results are bit-identical to the chain / fused2
// strategies.  Pass algebra: SURVEY.md 3.3 / pyfft/kernel.mako:805-1047 (two global passes).
// Safety: every spin is bounded and sticky (one time-out sets the error word, every later wait of every work-group
// returns at once); a census that does not find exactly 64 resident work-groups per XCD aborts before any data moves.
#pragma once
#include "fft_col2.hpp"

namespace mifft {

struct Xcd2Args {
    TileArgs p0;     // pass 0: in0/in1 = user input, tw_L = w(1024), tw_lo/tw_hi/tw_shift = w(2^20) two-level table
    TileArgs p1;     // pass 1: out0/out1 = user output, tw_L = w(1024), scale
    unsigned* ctl;   // control block, zeroed per launch: [0] arrivals [1] error [8..16) work-groups per XCD
                     //   ready[x][64] then rdone[x][64], one flag per 128-byte line (kXcd2FS words apart)
    void* scratch;   // [8 XCDs][64 consumers][16 slots][256 threads] complex<float>: 2 MiB per XCD
    unsigned batch;
    unsigned long long* trace;   // development: 32 time stamps (100 MHz) per work-group for transform index trace_iter, or null
    unsigned trace_iter;
    unsigned pace;               // development: throttle the HBM burst of the last stage
    unsigned phase_us;           // development (round 5): XCDs with an odd HW_REG_XCC_ID start this many microseconds late, so that about
                                 // half of the XCDs are in their HBM burst while the other half exchange (profiles/r05_xcd2_antiphase.log)
};
constexpr unsigned kXcd2FS = 32u;   // words between two flags: one 128-byte line per flag (packed, the 64 flags of an XCD shared two lines
                                    // that all its work-groups poll)
constexpr int kXcd2CtlWords = 64 + 2 * 512 * (int)kXcd2FS;
constexpr unsigned kXcd2ErrTimeout = 1u, kXcd2ErrCensus = 2u;

__device__ __forceinline__ unsigned xcd2_xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7u;
}

// Wave 0 waits until the 32 flag words flags(lane)[idx(lane)] (lanes 0..31; per-lane array, index and target) are all >= target.  Flags live in this XCD's
// L2: written with plain (workgroup-scope) stores by work-groups of the same XCD, polled with sc1 (L1-bypassing)
// loads.  Returns false after a time-out or when another work-group has already raised the error word.
__device__ __forceinline__ bool xcd2_wait16(const unsigned* flags, unsigned idx, unsigned target, unsigned* err) {
    const bool active = (threadIdx.x & 63u) < 16u;
    for (unsigned spins = 0;; ++spins) {
        bool ok = true;
        if (active) ok = __hip_atomic_load(flags + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target;
        if (__all(ok)) return true;
        __builtin_amdgcn_s_sleep(1);
        if ((spins & 255u) == 255u) {
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
            if (spins > (1u << 19)) {
                if ((threadIdx.x & 63u) == 0u) __hip_atomic_fetch_or(err, kXcd2ErrTimeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
    }
}

// stage 1 of one first-stage butterfly ia: radix-16 over b1, twiddle w(1024)^(b0*qb1) * w(64)^(ia*qb1)
template <typename T, int ia> __device__ __forceinline__ void xcd2_stage1(cplx<T>* v, const ColStageTw<T>& tw) {
    Dft<16, T>::run(v + ia * 16);
    static_for<15>([&](auto q2) {
        constexpr int qb1 = q2 + 1;
        cplx<T> t = v[ia * 16 + qb1];
        if constexpr (ia > 0) t = mul_w64<ia * qb1, T>(t);
        v[ia * 16 + qb1] = cmul<T>(t, tw.template get<qb1>());
    });
    __builtin_amdgcn_sched_barrier(0);
}

// Materialise a value at this program point (no instruction): LLVM otherwise SINKS the stage-2 butterflies of the slabs used
// in later rounds past the round-0 hand-off, where their 64 inputs stay live next to the arriving data (385 spills).
template <typename T> __device__ __forceinline__ void xcd2_pin(cplx<T>& z) { asm volatile("" : "+v"(z.x), "+v"(z.y)); }

// stage 2: radix-4 over a (16 butterflies), twiddle w(64)^(b0*qa) = w(1024)^(16*b0*qa)
template <typename T> __device__ __forceinline__ void xcd2_stage2(cplx<T>* v, const cplx<T>* twL, int b0) {
    cplx<T> twA[3];
    static_for<3>([&](auto qq) {
        constexpr int qa = qq + 1;
        twA[qq] = twL[16 * b0 * qa];
    });
    static_for<16>([&](auto bb) {
        constexpr int qb1 = bb;
        cplx<T> t[4];
        static_for<4>([&](auto aa) {
            constexpr int ia = aa;
            t[ia] = v[ia * 16 + qb1];
        });
        Dft<4, T>::run(t);
        v[qb1] = t[0];
        static_for<3>([&](auto qq) {
            constexpr int qa = qq + 1;
            v[qa * 16 + qb1] = cmul<T>(t[qa], twA[qq]);
        });
    });
    static_for<64>([&](auto kk) { xcd2_pin<T>(v[kk]); });
    __builtin_amdgcn_sched_barrier(0);
}

// phase-1 loads of pass 0 for transform t: v[ia*16 + b1] = in[t][(b1*64 + ia*16 + b0) * 1024 + rem0 + c]; slab ia only
template <typename T, bool SPLIT, bool NT, int ia>
__device__ __forceinline__ void xcd2_load_slab(const TileArgs& a, long long t, long long rem0, unsigned voff, int sh, cplx<T>* v) {
    // sh = 10 held in an SGPR the compiler cannot see through: the row offsets then are scalar arithmetic per access and the
    // loads take the (SGPR base + 32-bit VGPR offset) form instead of 64 loop-invariant VGPR address pairs (which spill)
    const long long ubase = t * a.ostride_in + rem0;
    if constexpr (!SPLIT) {
        const char* src = reinterpret_cast<const char*>(reinterpret_cast<const cplx<T>*>(a.in0) + ubase);
        const unsigned vb = voff * (unsigned)sizeof(cplx<T>);
        static_for<16>([&](auto bb) {
            constexpr int b1 = bb;
            const char* p = src + ((long long)(b1 * 64 + ia * 16) << sh) * (long long)sizeof(cplx<T>);
            if constexpr (NT) v[ia * 16 + b1] = __builtin_nontemporal_load(reinterpret_cast<const cplx<T>*>(p + vb));
            else v[ia * 16 + b1] = *reinterpret_cast<const cplx<T>*>(p + vb);
        });
    } else {
        const char* sre = reinterpret_cast<const char*>(reinterpret_cast<const T*>(a.in0) + ubase);
        const char* sim = reinterpret_cast<const char*>(reinterpret_cast<const T*>(a.in1) + ubase);
        const unsigned vb = voff * (unsigned)sizeof(T);
        static_for<16>([&](auto bb) {
            constexpr int b1 = bb;
            const long long off = ((long long)(b1 * 64 + ia * 16) << sh) * (long long)sizeof(T);
            v[ia * 16 + b1].x = *reinterpret_cast<const T*>(sre + off + vb);
            v[ia * 16 + b1].y = *reinterpret_cast<const T*>(sim + off + vb);
        });
    }
}

// MODE (development, elimination measurements of profiles/r03_xcd2_elimination.log; every mode but 0 gives WRONG results):
//   0 the kernel   1 no exchange waits (flags still written)   2 no exchange at all (no scratch traffic, no flags)
//   3 HBM only (loads and stores, no butterflies, no LDS, no exchange)   5 butterflies and LDS only
//   (4 = no HBM but exchange with waits: compiles to 750-1950 spilled registers without the loads' live ranges -- not instantiated)
template <typename T, int ROT, bool SPLIT, bool NT, bool PREFETCH, int MODE = 0>
__device__ __forceinline__ void xcd2_body(const Xcd2Args& f, const unsigned x, const unsigned r, cplx<T>* lds, unsigned* s_ok) {
    static_assert(sizeof(cplx<T>) == 8, "fp32 only: the hand-off moves one 8-byte word per point");
    constexpr bool kNoWait = MODE == 1, kNoXchg = MODE == 2 || MODE == 3 || MODE == 5, kNoFlop = MODE == 3,
                   kNoHbm = MODE == 4 || MODE == 5;
    auto load_all = [&](long long t, unsigned voff, int sh, cplx<T>* v) __attribute__((always_inline)) {
        if constexpr (kNoHbm) static_for<64>([&](auto kk) { v[kk] = cplx<T>{(T)(voff & 1023u), (T)(int)(kk & 3)}; });
        else static_for<4>([&](auto aa) { xcd2_load_slab<T, SPLIT, NT, aa>(f.p0, t, (long long)r * 16, voff, sh, v); });
    };
    cplx<T> sink = {(T)0, (T)0};
    constexpr int PITCH0 = 17;                 // pass 0 (transposing) exchange: [b0][c][qb1 + pad]
    constexpr int BUF1 = 16 * 16 * 16;         // pass 1 exchange: [b0][qb1][c], double-buffered
    const int tid0 = threadIdx.x;
    const bool wave0 = tid0 < 64;
    const unsigned lane16 = (unsigned)tid0 & 15u;

    unsigned* const err = f.ctl + 1;
    unsigned* const ready = f.ctl + 64 + 64 * x * kXcd2FS;
    unsigned* const rdone = f.ctl + 64 + (512 + 64 * x) * kXcd2FS;
    char* const sbase = reinterpret_cast<char*>(f.scratch) + (size_t)x * (64u * 16u * 256u * 8u);
    const unsigned slot = r >> 2;              // b1 of my columns in every consumer's first-stage butterfly
    const long long rem0 = (long long)r * 16;  // my 16 columns, both passes

    const cplx<T>* twL0 = reinterpret_cast<const cplx<T>*>(f.p0.tw_L);
    const cplx<T>* twL1 = reinterpret_cast<const cplx<T>*>(f.p1.tw_L);
    const cplx<T>* twlo = reinterpret_cast<const cplx<T>*>(f.p0.tw_lo);
    const cplx<T>* twhi = reinterpret_cast<const cplx<T>*>(f.p0.tw_hi);
    const int tw_shift = f.p0.tw_shift;
    const unsigned lomask = (1u << tw_shift) - 1u;
    const bool inverse = f.p0.inverse != 0;
    const T csign = inverse ? (T)-1 : (T)1;
    const T sx = (T)f.p1.scale;
    const T sy = inverse ? -sx : sx;
    const unsigned voff_in0 = (((unsigned)tid0 >> 4) << 10) + ((unsigned)tid0 & 15u);   // (b0 << logMS) + c

    const unsigned nx = (f.batch + 7u - x) >> 3;   // this XCD's transforms: t = x + 8 i
    cplx<T> v[64];
    if (nx > 0) load_all((long long)x, voff_in0, 10, v);
    bool alive = true;
    for (unsigned i = 0; i < nx && alive; ++i) {
        const bool tracing = f.trace != nullptr && i == f.trace_iter;
        auto stamp = [&](int idx) {
            if (tracing && threadIdx.x == 0) f.trace[(size_t)(x * 64u + r) * 32u + (unsigned)idx] = wall_clock64();
        };
        stamp(0);
        const long long t = (long long)x + 8ll * i;
        const unsigned g = 4u * i;   // global round number of this transform's round 0
        // loop-invariant scalars laundered once per transform: every address below is then (scalar base computed at the
        // access) + (one 32-bit VGPR offset), not a loop-invariant 64-bit VGPR pair hoisted out of the loop and spilled
        int sh10 = 10;
        unsigned slot_l = slot, r_l = r;
        asm volatile("" : "+s"(sh10), "+s"(slot_l), "+s"(r_l));
        // likewise the thread index: everything derived from it (twiddle addresses and the table twiddles themselves, ~130
        // registers' worth) is recomputed per transform instead of living across the whole loop
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const int lo4 = tid & 15, hi4 = tid >> 4;  // pass 0 phase 1: (c, b0); pass 0 phase 2: (u, c2); pass 1: (c, b0) and (c2, u)
        const unsigned voff_in = ((unsigned)hi4 << sh10) + (unsigned)lo4;
        // ================= pass 0: stages 1 and 2 on the loaded columns (thread = (c = lo4, b0 = hi4))
        if constexpr (!kNoFlop) {
            static_for<64>([&](auto kk) {
                constexpr int k = kk;
                v[k].y *= csign;
            });
            {
                ColStageTw<T> tw;
                tw.init(twL0, hi4);
                static_for<4>([&](auto aa) { xcd2_stage1<T, aa>(v, tw); });
            }
            xcd2_stage2<T>(v, twL0, hi4);
        }
        stamp(1);
        // ================= four rounds: slab qa of pass 0 out, first-stage butterfly ia of pass 1 in.  The butterflies of
        // the next slab run under the store acknowledgements, stage 1 of the previous arrival under the load latency:
        //     compute(0) stores(0) | compute(1) A(0) loads(0) B(0) stores(1) | compute(2) A(1) loads(1) st1(y0) B(1) stores(2) | ...
        cplx<T> y[64];
        cplx<T> xs[16];   // the slab that has been computed but not yet stored
        auto compute = [&](auto kk) {
            constexpr int k = kk;
            constexpr int qa = (ROT - k) & 3;    // slab I send in round k; my consumers are r' = qb0*4 + qa
            // pass 0 phase 2 (thread = (u = lo4, c2 = hi4)): LDS exchange of slab qa, radix-16, inter-pass twiddle
            if constexpr (k > 0) __syncthreads();   // the previous round's LDS reads are over
            static_for<16>([&](auto ss) {
                constexpr int qb1 = ss;
                lds[(hi4 * 16 + lo4) * PITCH0 + qb1] = v[qa * 16 + qb1];
            });
            __syncthreads();
            static_for<16>([&](auto bb) {
                constexpr int bi = bb;
                xs[bi] = lds[(bi * 16 + hi4) * PITCH0 + lo4];
            });
            __builtin_amdgcn_sched_barrier(0);
            Dft<16, T>::run(xs);
            __builtin_amdgcn_sched_barrier(0);   // the twiddle look-ups start after the butterfly's temporaries are gone
            {
                const unsigned l = (unsigned)rem0 + (unsigned)hi4;
                auto look = [&](unsigned e) { return cmul<T>(twlo[e & lomask], twhi[e >> tw_shift]); };
                const cplx<T> sstep = look(l * 64u);
                static_for<4>([&](auto jj) {
                    constexpr int j = jj;
                    cplx<T> cur = look(l * (unsigned)(qa * 16 + lo4 + 256 * j));
                    static_for<4>([&](auto ii) {
                        constexpr int qb0 = 4 * j + ii;
                        xs[qb0] = cmul<T>(xs[qb0], cur);
                        if constexpr (ii < 3) cur = cmul<T>(cur, sstep);
                    });
                });
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        auto stores = [&](auto kk) {
            constexpr int k = kk;
            constexpr int qa = (ROT - k) & 3;
            constexpr unsigned par = 0u;
            if constexpr (kNoXchg) return;
            static_for<16>([&](auto qq) {
                constexpr int qb0 = qq;
                char* p = sbase + ((size_t)((par * 64u + (unsigned)(qb0 * 4 + qa)) * 16u + slot_l) << (sh10 + 1));
                // the pass-0 store conjugates for the inverse transform and pass 1 conjugates again on load: both folded away
                *reinterpret_cast<cplx<T>*>(p + (unsigned)tid * 8u) = xs[qb0];
            });
        };
        auto loads = [&](auto kk) {
            constexpr int k = kk;
            constexpr int ia = (ROT + k) & 3;    // butterfly I receive in round k; my producers are r = j*4 + ia
            constexpr unsigned par = 0u;
            if constexpr (kNoXchg) {
                static_for<16>([&](auto jj) { y[ia * 16 + jj] = xs[jj]; });
                return;
            }
            static_for<16>([&](auto jj) {
                constexpr int j = jj;
                const char* p = sbase + ((size_t)((par * 64u + r_l) * 16u + (unsigned)j) << (sh10 + 1));
                const unsigned long long w = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p + (unsigned)tid * 8u),
                                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                y[ia * 16 + j] = __builtin_bit_cast(cplx<T>, w);
            });
        };
        // One 2 MiB buffer per XCD (what stays resident in the 4 MiB L2 next to the streams: profiles/r02_e_*), so every
        // round has TWO rendezvous: A = my 16 producers have published round k (then loads(k)); B = my 16 consumers of
        // round k+1 have read round k (then stores(k+1) may overwrite their blocks).
        auto rendezvous = [&](unsigned* mine, unsigned value, const unsigned* theirs, unsigned sel) {
            if constexpr (kNoXchg) return;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // A: stores(k) acknowledged; B: loads(k) landed
            __syncthreads();
            if (tid == 0) __hip_atomic_store(mine + r * kXcd2FS, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (wave0) {
                const bool ok = kNoWait ? true : xcd2_wait16(theirs, (lane16 * 4u + sel) * kXcd2FS, value, err);
                if (tid == 0) *s_ok = ok ? 1u : 0u;
            }
            __syncthreads();
            alive = alive && (__builtin_amdgcn_readfirstlane(*s_ok) != 0u);
        };
        auto stage1_y = [&](auto kk) {
            constexpr int ia = (ROT + kk) & 3;
            // pass 1 stage 1 (thread = (c = lo4, b0 = hi4)); the table twiddles are looked up again per butterfly (L1 hits)
            // rather than held across the rounds (16 registers)
            ColStageTw<T> tw1;
            tw1.init(twL1, hi4);
            xcd2_stage1<T, ia>(y, tw1);
        };
        stamp(2);
        if constexpr (kNoFlop) {
            static_for<64>([&](auto kk) { y[kk] = v[kk]; });
        } else {
        compute(IC<0>{});
        stores(IC<0>{});       // (my consumers of round 0 have read the previous transform: rendezvous B of its round 3)
        static_for<4>([&](auto kk) {
            constexpr int k = kk;
            if constexpr (k < 3) compute(IC<k + 1>{});          // under the store acknowledgements
            rendezvous(ready, g + k + 1u, ready, (unsigned)((ROT + k) & 3));
            stamp(3 + 2 * k);
            loads(IC<k>{});
            if constexpr (k > 0) stage1_y(IC<k - 1>{});         // under the load latency
            rendezvous(rdone, g + k + 1u, rdone, (unsigned)((ROT - k - 1) & 3));
            stamp(4 + 2 * k);
            if constexpr (k < 3) stores(IC<k + 1>{});
        });
        stage1_y(IC<3>{});
        if (!alive) break;
        xcd2_stage2<T>(y, twL1, hi4);
        }
        stamp(11);

        // ================= pass 1 phase 2 (thread = (c2 = lo4, u = hi4)): exchange, radix-16, store; prefetch the next transform
        const bool more = PREFETCH && (i + 1u < nx);
        const long long oubase = f.p1.ostride_out * t + rem0;
        const unsigned ovoff = ((unsigned)hi4 << 10) + (unsigned)lo4;
        __syncthreads();   // pass 0's last LDS reads are over
        static_for<4>([&](auto rr) {
            constexpr int qa = rr;
            cplx<T>* buf = lds + (qa & 1) * BUF1;
            cplx<T> xv[16];
            if constexpr (kNoFlop) {
                static_for<16>([&](auto ss) { xv[ss] = y[qa * 16 + ss]; });
            } else {
            static_for<16>([&](auto ss) {
                constexpr int qb1 = ss;
                buf[(hi4 * 16 + qb1) * 16 + lo4] = y[qa * 16 + qb1];
            });
            __syncthreads();
            static_for<16>([&](auto bb) {
                constexpr int bi = bb;
                xv[bi] = buf[(bi * 16 + hi4) * 16 + lo4];
            });
            Dft<16, T>::run(xv);
            }
            static_for<16>([&](auto qq) {
                constexpr int qb0 = qq;
                const long long gu = oubase + ((long long)(qb0 * 64 + 16 * qa) << sh10);
                cplx<T> o;
                o.x = xv[qb0].x * sx;
                o.y = xv[qb0].y * sy;
                if constexpr (kNoHbm) {
                    sink += o;
                } else if constexpr (!SPLIT) {
                    char* p = reinterpret_cast<char*>(reinterpret_cast<cplx<T>*>(f.p1.out0) + gu);
                    if constexpr (NT) __builtin_nontemporal_store(o, reinterpret_cast<cplx<T>*>(p + ovoff * (unsigned)sizeof(cplx<T>)));
                    else *reinterpret_cast<cplx<T>*>(p + ovoff * (unsigned)sizeof(cplx<T>)) = o;
                } else {
                    char* pr = reinterpret_cast<char*>(reinterpret_cast<T*>(f.p1.out0) + gu);
                    char* pi = reinterpret_cast<char*>(reinterpret_cast<T*>(f.p1.out1) + gu);
                    *reinterpret_cast<T*>(pr + ovoff * (unsigned)sizeof(T)) = o.x;
                    *reinterpret_cast<T*>(pi + ovoff * (unsigned)sizeof(T)) = o.y;
                }
            });
            if (f.pace) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");   // development: self-throttled HBM burst
            if constexpr (PREFETCH) {
                // (the else branch ends the live range of the consumed slab: without it the stale registers stay live
                // around the whole loop through the PHI of the conditional load and the kernel spills)
                if constexpr (kNoHbm) {
                    static_for<16>([&](auto bb) { v[qa * 16 + bb] = sink; });   // (the next transform's "data": nothing to hold)
                } else {
                if (more) xcd2_load_slab<T, SPLIT, NT, qa>(f.p0, t + 8, rem0, voff_in, sh10, v);
                else static_for<16>([&](auto bb) { v[qa * 16 + bb] = cplx<T>{(T)0, (T)0}; });
                }
            }
        });
        if constexpr (!PREFETCH) {
            if (i + 1u < nx) static_for<4>([&](auto aa) { xcd2_load_slab<T, SPLIT, NT, aa>(f.p0, t + 8, rem0, voff_in, sh10, v); });
            else static_for<64>([&](auto kk) { v[kk] = cplx<T>{(T)0, (T)0}; });
        }
        stamp(12);
        __syncthreads();   // LDS is free for the next transform's pass 0
    }
    if constexpr (kNoHbm) {   // (keeps the arithmetic alive)
        if (sink.x == (T)123456.789) *reinterpret_cast<cplx<T>*>(f.p1.out0) = sink;
    }
}

template <typename T, bool SPLIT, bool NT, bool PREFETCH, int MODE = 0>
__global__ void __launch_bounds__(256, 2) fft_xcd2_kernel(const Xcd2Args f) {
    __shared__ __attribute__((aligned(16))) cplx<T> lds[2 * 16 * 16 * 16];   // 64 KiB: pass 0 uses 34 KiB of it
    __shared__ unsigned s_w[4];
    const unsigned x = xcd2_xcc_id();
    unsigned* const err = f.ctl + 1;
    if (threadIdx.x == 0) {
        // census: my rank on this XCD, then wait for the whole grid so that the per-XCD counts are final
        const unsigned r = __hip_atomic_fetch_add(f.ctl + 8 + x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(f.ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned ok = 1u;
        for (unsigned spins = 0; __hip_atomic_load(f.ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x; ++spins) {
            __builtin_amdgcn_s_sleep(8);
            if (spins > (1u << 20) || ((spins & 63u) == 63u && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                ok = 0u;
                break;
            }
        }
        if (ok && __hip_atomic_load(f.ctl + 8 + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 64u) ok = 0u;
        if (!ok) __hip_atomic_fetch_or(err, kXcd2ErrCensus, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_w[0] = r;
        s_w[1] = ok;
    }
    __syncthreads();
    const unsigned r = __builtin_amdgcn_readfirstlane(s_w[0]);   // (an LDS read is not provably uniform: without this every address is a VGPR pair)
    if (s_w[1] == 0u || r >= 64u) return;
    if (f.phase_us != 0u && (x & 1u)) {   // anti-phase start of the odd XCDs (100 MHz wall clock)
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < (unsigned long long)f.phase_us * 100ull) __builtin_amdgcn_s_sleep(32);
    }
    switch (r & 3u) {
        case 0: xcd2_body<T, 0, SPLIT, NT, PREFETCH, MODE>(f, x, r, lds, s_w + 2); break;
        case 1: xcd2_body<T, 1, SPLIT, NT, PREFETCH, MODE>(f, x, r, lds, s_w + 2); break;
        case 2: xcd2_body<T, 2, SPLIT, NT, PREFETCH, MODE>(f, x, r, lds, s_w + 2); break;
        default: xcd2_body<T, 3, SPLIT, NT, PREFETCH, MODE>(f, x, r, lds, s_w + 2); break;
    }
}

}  // namespace mifft

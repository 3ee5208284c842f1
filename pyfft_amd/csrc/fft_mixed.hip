// Mixed-radix rows and lines: transforms of SMOOTH length n = 2^a 3^b 5^c 7^d (the reference's TODO.txt:8, "support for
// non-power-of-2 sized arrays"), one Stockham transform per row with radix-3 / 5 / 7 butterflies and their composites (6, 9, 10, 12,
// 14, 15) next to the power-of-two ones -- instead of Bluestein's three padded power-of-two transforms and four streaming copies
// (pyfft_amd/generic.py), which stays for lengths with a larger prime factor.
//
// A work-group owns W transforms of n points (W * n <= 2048 fp32 / 1024 fp64 points when a transform fits that, two LDS buffers of
// that size: 32 KiB, four work-groups per CU).  The first stage takes its operands from HBM, the last one stores to HBM, the stages
// in between read one LDS buffer and write the other at the autosort position:
//     stage with radix R, Ns = product of the earlier radices:  butterfly jb < n / R
//         reads   x[jb + k * n / R]                                   k < R
//         twiddle w(n)^(k * (jb mod Ns) * n / (Ns * R))               (table of n entries, float64-evaluated on the host)
//         writes  y[(jb div Ns) * Ns * R + (jb mod Ns) + k * Ns]
// (the same algebra as fft_tile.hpp, without the power-of-two shortcuts: lengths are run-time values, the radix list comes with
// the launch; the radix is a compile-time constant inside each stage loop).  The inverse is conj -> forward -> conj.  Interleaved
// data, in place or out of place; rows `stride` apart, or the lines of a strided axis (`inner` elements between a line's points).
#include "fft_mixed.hpp"

namespace {

struct MixedArgs {
    const void* in;
    void* out;
    const void* tw;        // n entries w(n)^m
    const void* tw_lo;     // long transforms (mode 2): w(N)^e = tw_lo[e & (2^tw_shift - 1)] * tw_hi[e >> tw_shift], N = n * inner
    const void* tw_hi;
    int tw_shift;
    long long rows, stride_in, stride_out;
    long long inner;       // 1: contiguous rows `stride` apart.  > 1: LINES of a strided axis -- line L = o * inner + j starts at element
                           // o * n * inner + j and its points are `inner` elements apart (stride_in / stride_out unused)
    int n, W, nstages, inverse;
    int conj_in, conj_out; // conjugate on load / on store (inverse = both; an N-D plan conjugates once at either end)
    int radix[kMaxStages];
    // 1 / (n / R) and 1 / Ns per stage: index divisions as one float multiply (indices < 2^22, exact with the + 0.5 below)
    float inv_per_row[kMaxStages], inv_ns[kMaxStages];
    double scale;
};

// one butterfly of radix R, transform-local index jb: operands from `src` (twiddled), DFT in registers, results to `dst` at the
// autosort position.  A transform's point i sits at base + i * es: es = 1 for ROWS (the tile's rows one after the other), es = the
// number of the tile's lines for LINES (the tile is [point][line] in LDS, and `inner` elements apart in HBM: adjacent lanes take
// adjacent lines, so both sides are coalesced).  GIN / GOUT: `src` / `dst` is HBM (the first / last stage: consecutive butterflies
// read consecutive points of every operand, and the last stage, Ns = n / R, writes consecutive points of every result), with the
// conjugation of the inverse direction / the scale folded in.
template <int R, typename T, bool GIN, bool GOUT, bool SL, bool DL, bool BIGTW, typename SI, typename DI>
__device__ __forceinline__ void stage_butterfly(const cplx<T>* src, SI es_s, cplx<T>* dst, DI es_d, const cplx<T>* tw, int LR, int Ns,
                                                float inv_ns, int jb, T csign, T sx, T sy, const cplx<T>* twlo, const cplx<T>* twhi,
                                                int tw_shift, unsigned line) {
    const int jm = jb - fast_div(jb, inv_ns) * Ns;
    cplx<T> v[R];
    static_for<R>([&](auto kk) {
        if constexpr (SL) v[kk] = src[(SI)(jb + kk * LR) * es_s];
        else v[kk] = src[jb + kk * LR];
    });
    if constexpr (GIN) static_for<R>([&](auto kk) { v[kk].y *= csign; });
    if (Ns > 1) {
        const int step = jm * (LR / Ns);                 // jm * n / (Ns * R)
        static_for<R - 1>([&](auto kk) {
            constexpr int k = kk + 1;
            v[k] = cmul<T>(v[k], tw[k * step]);
        });
    }
    dft_any<R, T>(v);
    const int q0 = (jb - jm) * R + jm;
    static_for<R>([&](auto kk) {
        cplx<T> p = v[kk];
        if constexpr (BIGTW) {
            const unsigned e = (unsigned)(q0 + kk * Ns) * line;           // < N <= 2^24
            p = cmul<T>(p, cmul<T>(twlo[e & ((1u << tw_shift) - 1u)], twhi[e >> tw_shift]));
        }
        if constexpr (GOUT) {
            p.x *= sx;
            p.y *= sy;
        }
        if constexpr (DL) dst[(DI)(q0 + kk * Ns) * es_d] = p;
        else dst[q0 + kk * Ns] = p;
    });
}

struct StageCtx {
    const void* gin;       // HBM: first transform of the tile (rows) / first line of the tile (lines)
    void* gout;
    long long stride_in, stride_out, inner;
    const void* tw_lo;
    const void* tw_hi;
    int tw_shift, line0;   // mode 2: the tile's first line within its array
    int n, nrows, es;      // es: LDS distance between a line's points (lines: the number of lines; mode 2: that made odd)
    float inv_rows;
};

// one whole stage of the tile: butterflies tid, tid + NT, ... (the radix is a compile-time constant INSIDE the loop, so the
// compiler overlaps the operand loads of consecutive butterflies).  MODE 0: rows.  1: lines.  2: lines in, ROWS out (the first
// pass of a long transform N = n * inner: the W lines of a tile leave as W contiguous rows of n points, times w(N)^(q * line)):
// every stage but the last is the lines form; the last one walks the tile row by row (consecutive lanes, consecutive q).
template <int R, typename T, int NT, bool GIN, bool GOUT, int MODE>
__device__ __forceinline__ void run_stage(const StageCtx& c, const cplx<T>* src, cplx<T>* dst, const cplx<T>* tw, int Ns, float ipr, float ins,
                                          T csign, T sx, T sy) {
    constexpr bool LINES = MODE != 0;
    constexpr bool XOUT = MODE == 2 && GOUT;             // rows out of a lines tile
    const int n = c.n, per_row = n / R, total = c.nrows * per_row;
    for (int j = threadIdx.x; j < total; j += NT) {
        int r, jb;
        if constexpr (LINES && !XOUT) { jb = fast_div(j, c.inv_rows); r = j - jb * c.nrows; }
        else { r = fast_div(j, ipr); jb = j - r * per_row; }
        const cplx<T>* sr;
        cplx<T>* dr;
        if constexpr (GIN) sr = reinterpret_cast<const cplx<T>*>(c.gin) + (LINES ? (long long)r : r * c.stride_in);
        else sr = src + (LINES ? r : r * n);
        if constexpr (XOUT) dr = reinterpret_cast<cplx<T>*>(c.gout) + (long long)r * n;
        else if constexpr (GOUT) dr = reinterpret_cast<cplx<T>*>(c.gout) + (LINES ? (long long)r : r * c.stride_out);
        else dr = dst + (LINES ? r : r * n);
        using SI = typename std::conditional<GIN, long long, int>::type;
        using DI = typename std::conditional<GOUT, long long, int>::type;
        stage_butterfly<R, T, GIN, GOUT, LINES, LINES && !XOUT, XOUT, SI, DI>(
            sr, GIN ? (SI)c.inner : (SI)c.es, dr, GOUT ? (DI)c.inner : (DI)c.es, tw, per_row, Ns, ins, jb, csign, sx, sy,
            reinterpret_cast<const cplx<T>*>(c.tw_lo), reinterpret_cast<const cplx<T>*>(c.tw_hi), c.tw_shift, (unsigned)(c.line0 + r));
    }
}

template <typename T, int NT, bool GIN, bool GOUT, int MODE, typename... Args>
__device__ __forceinline__ void stage_switch(int R, Args&&... args) {
    switch (R) {
        case 2: run_stage<2, T, NT, GIN, GOUT, MODE>(args...); break;
        case 3: run_stage<3, T, NT, GIN, GOUT, MODE>(args...); break;
        case 4: run_stage<4, T, NT, GIN, GOUT, MODE>(args...); break;
        case 5: run_stage<5, T, NT, GIN, GOUT, MODE>(args...); break;
        case 6: run_stage<6, T, NT, GIN, GOUT, MODE>(args...); break;
        case 7: run_stage<7, T, NT, GIN, GOUT, MODE>(args...); break;
        case 8: run_stage<8, T, NT, GIN, GOUT, MODE>(args...); break;
        case 9: run_stage<9, T, NT, GIN, GOUT, MODE>(args...); break;
        case 10: run_stage<10, T, NT, GIN, GOUT, MODE>(args...); break;
        case 12: run_stage<12, T, NT, GIN, GOUT, MODE>(args...); break;
        case 14: run_stage<14, T, NT, GIN, GOUT, MODE>(args...); break;
        case 15: run_stage<15, T, NT, GIN, GOUT, MODE>(args...); break;
        default: run_stage<16, T, NT, GIN, GOUT, MODE>(args...); break;
    }
}

// Two LDS buffers of W * n points: a stage reads one and writes the other (one barrier per stage); the first stage reads HBM and the
// last one writes it (N = 1000 fp32: 32.5 % of the roofline through staging copies, 49.4 % with these register edges).
template <typename T, int NT, int MODE, int OCC>
__global__ void __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(OCC))) fft_mixed_kernel(const MixedArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cplx<T>* lds = reinterpret_cast<cplx<T>*>(smem);
    const cplx<T>* tw = reinterpret_cast<const cplx<T>*>(a.tw);
    const int n = a.n, W = a.W;
    const long long row0 = (long long)blockIdx.x * W;
    StageCtx c;
    c.n = n;
    c.nrows = (int)((a.rows - row0) < W ? (a.rows - row0) : W);
    c.stride_in = a.stride_in; c.stride_out = a.stride_out; c.inner = a.inner;
    c.inv_rows = 1.0f / (float)c.nrows;
    c.tw_lo = a.tw_lo; c.tw_hi = a.tw_hi; c.tw_shift = a.tw_shift; c.line0 = 0;
    c.es = c.nrows;
    int half = W * n;
    if constexpr (MODE != 0) {
        // (the launcher makes W divide `inner`, so a tile never straddles two values of o; its W lines are adjacent in memory)
        const long long o = row0 / a.inner, j0 = row0 - o * a.inner;
        const long long line_base = o * (long long)n * a.inner + j0;
        c.gin = reinterpret_cast<const cplx<T>*>(a.in) + line_base;
        if constexpr (MODE == 2) {
            c.gout = reinterpret_cast<cplx<T>*>(a.out) + (o * a.inner + j0) * (long long)n;     // line j0 + r -> row j0 + r of n points
            c.line0 = (int)j0;
            c.es = c.nrows | 1;        // the row-by-row walk of the last stage reads LDS `es` points apart: odd = conflict-free
            half = (W | 1) * n;
        } else {
            c.gout = reinterpret_cast<cplx<T>*>(a.out) + line_base;
        }
    } else {
        c.gin = reinterpret_cast<const cplx<T>*>(a.in) + row0 * a.stride_in;
        c.gout = reinterpret_cast<cplx<T>*>(a.out) + row0 * a.stride_out;
    }
    const T csign = a.conj_in ? (T)-1 : (T)1;
    const T sx = (T)a.scale, sy = a.conj_out ? -sx : sx;
    int Ns = 1, cur = 0;
    for (int s = 0; s < a.nstages; ++s) {
        const int R = a.radix[s];
        const float ipr = a.inv_per_row[s], ins = a.inv_ns[s];
        const cplx<T>* src = lds + cur * half;
        cplx<T>* dst = lds + (cur ^ 1) * half;
        const bool gin = s == 0, gout = s == a.nstages - 1;
        if (gin && gout) stage_switch<T, NT, true, true, MODE>(R, c, src, dst, tw, Ns, ipr, ins, csign, sx, sy);
        else if (gin) stage_switch<T, NT, true, false, MODE>(R, c, src, dst, tw, Ns, ipr, ins, csign, sx, sy);
        else if (gout) stage_switch<T, NT, false, true, MODE>(R, c, src, dst, tw, Ns, ipr, ins, csign, sx, sy);
        else stage_switch<T, NT, false, false, MODE>(R, c, src, dst, tw, Ns, ipr, ins, csign, sx, sy);
        if (!gout) __syncthreads();
        Ns *= R;
        cur ^= 1;
    }
}

}  // namespace

// 0 if rows of n points have a mixed-radix kernel: n = 2^a 3^b 5^c 7^d, 2 <= n <= 4096 (fp32) / 2048 (fp64)
extern "C" int mifft_mixed_supported_impl(int f64, int n) {
    int radix[kMaxStages];
    if (n < 2 || n > (f64 ? kTilePoints64 : kTilePoints32)) return -2;
    return factor(n, radix) ? 0 : -2;
}

namespace {

// lines per tile: the largest W <= cap / n that divides `inner` (a tile's W lines are adjacent and share one o); the full tile when
// the half one holds fewer than 16 lines (a line's points are W * sizeof(complex) bytes of one memory segment)
int lines_per_tile(int f64, int n, long long inner, bool odd_pad) {
    const int full = f64 ? kTilePoints64 : kTilePoints32;
    auto fit = [&](int cap) {
        int W = cap / n;
        if (odd_pad) while (W > 1 && (W | 1) * n > cap) --W;
        if (W < 1) W = 1;
        while (W > 1 && inner % W) --W;
        return W;
    };
    const int wh = n <= full / 2 ? fit(full / 2) : 0;
    return wh >= 16 ? wh : fit(full);
}

// mode 0: rows (inner == 1).  1: lines.  2: lines in, rows out, times w(n * inner)^(q * line) (tw_lo / tw_hi / tw_shift)
int mixed_launch_impl(int f64, int n, long long rows, long long stride_in, long long stride_out, long long inner, const void* in, void* out,
                      const void* tw, int flags, double scale, int mode, const void* tw_lo, const void* tw_hi, int tw_shift, hipStream_t s) {
    MixedArgs a;
    a.nstages = factor(n, a.radix);
    if (!a.nstages) return -2;
    a.in = in; a.out = out; a.tw = tw;
    a.tw_lo = tw_lo; a.tw_hi = tw_hi; a.tw_shift = tw_shift;
    a.rows = rows; a.stride_in = stride_in; a.stride_out = stride_out;
    a.n = n; a.inverse = (flags & 3) == 3; a.scale = scale;
    a.conj_in = flags & 1; a.conj_out = (flags >> 1) & 1;
    a.inner = inner < 1 ? 1 : inner;
    for (int i = 0, nsx = 1; i < a.nstages; ++i) {
        a.inv_per_row[i] = 1.0f / (float)(n / a.radix[i]);
        a.inv_ns[i] = 1.0f / (float)nsx;
        nsx *= a.radix[i];
    }
    int W;
    if (mode == 0) {
        // tiles of half the capacity (32 KiB of LDS, four work-groups per CU) whenever a row fits one: the kernel is latency-bound
        // (full tiles: N = 1000 fp32 39 % of the roofline, half: 49 %, quarter: 40 %)
        const int full = f64 ? kTilePoints64 : kTilePoints32;
        const int cap = full / (n <= full / 2 ? 2 : 1);
        W = cap / n;
        if (W < 1) W = 1;
        if (W > rows) W = (int)rows;
    } else {
        W = lines_per_tile(f64, n, a.inner, mode == 2);
    }
    a.W = W;
    const long long blocks = (rows + W - 1) / W;
    if (blocks <= 0) return 0;
    if (blocks > 2147483647ll) return -1;
    const size_t lds_bytes = 2 * (size_t)(mode == 2 ? (W | 1) : W) * n * (f64 ? 16 : 8);
    const dim3 g((unsigned)blocks), b(kThreads);
    if (f64) {
        if (mode == 2) hipLaunchKernelGGL((fft_mixed_kernel<double, kThreads, 2, OCC64>), g, b, lds_bytes, s, a);
        else if (mode == 1) hipLaunchKernelGGL((fft_mixed_kernel<double, kThreads, 1, OCC64>), g, b, lds_bytes, s, a);
        else hipLaunchKernelGGL((fft_mixed_kernel<double, kThreads, 0, OCC64>), g, b, lds_bytes, s, a);
    } else {
        if (mode == 2) hipLaunchKernelGGL((fft_mixed_kernel<float, kThreads, 2, OCC32>), g, b, lds_bytes, s, a);
        else if (mode == 1) hipLaunchKernelGGL((fft_mixed_kernel<float, kThreads, 1, OCC32>), g, b, lds_bytes, s, a);
        else hipLaunchKernelGGL((fft_mixed_kernel<float, kThreads, 0, OCC32>), g, b, lds_bytes, s, a);
    }
    return (int)hipGetLastError();
}

}  // namespace

// ---- Bluestein's algorithm in ONE launch -------------------------------------------------------------------------------------
// Rows of ANY length n whose padded length m >= 2 n - 1 (smooth, chosen below) fits a tile: with the chirp c[j] = exp(-i pi j^2 / n)
//     X[k] = c[k] * sum_j (x[j] c[j]) conj(c[k - j])  =  c[k] * IFFT_m( FFT_m(x c, zero-padded) * bhat )[k],   bhat = FFT_m(conj c, wrapped)
// and both transforms of m points are the stage loop above inside LDS: the first stage of the first transform reads HBM (times the
// chirp, zeros beyond n), its last stage multiplies by bhat / m and conjugates (the second transform is conj -> forward -> conj),
// the last stage of the second one multiplies by the chirp and stores the n results.  One HBM round trip instead of the three
// transforms and four streaming copies of pyfft_amd/generic.py.
namespace {

struct BlueArgs {
    const void* in;
    void* out;
    const void* tw;        // m entries w(m)^j
    const void* chirp;     // n entries
    const void* bhat;      // m entries, already divided by m
    long long rows, stride_in, stride_out;
    int n, m, W, nstages;
    int conj_in, conj_out;
    int radix[kMaxStages];
    float inv_per_row[kMaxStages], inv_ns[kMaxStages];
    double scale;
};

// PHASE 0: HBM -> LDS (first transform, first stage).  1: LDS -> LDS.  2: LDS -> LDS, times bhat, conjugated (first transform, last
// stage).  3: LDS -> HBM (second transform, last stage).
template <int R, typename T, int NT, int PHASE>
__device__ __forceinline__ void blue_stage(const BlueArgs& a, long long row0, int nrows, const cplx<T>* src, cplx<T>* dst, int Ns, float ipr,
                                           float ins) {
    const int n = a.n, m = a.m, LR = m / R, total = nrows * LR;
    const cplx<T>* tw = reinterpret_cast<const cplx<T>*>(a.tw);
    const cplx<T>* chirp = reinterpret_cast<const cplx<T>*>(a.chirp);
    for (int j = threadIdx.x; j < total; j += NT) {
        const int r = fast_div(j, ipr), jb = j - r * LR;
        const int jm = jb - fast_div(jb, ins) * Ns;
        cplx<T> v[R];
        if constexpr (PHASE == 0) {
            const cplx<T>* x = reinterpret_cast<const cplx<T>*>(a.in) + (row0 + r) * a.stride_in;
            const T cs = a.conj_in ? (T)-1 : (T)1;
            static_for<R>([&](auto kk) {
                const int i = jb + kk * LR;
                cplx<T> p = {(T)0, (T)0};
                if (i < n) {
                    p = x[i];
                    p.y *= cs;
                    p = cmul<T>(p, chirp[i]);
                }
                v[kk] = p;
            });
        } else {
            static_for<R>([&](auto kk) { v[kk] = src[r * m + jb + kk * LR]; });
        }
        if (Ns > 1) {
            const int step = jm * (LR / Ns);
            static_for<R - 1>([&](auto kk) {
                constexpr int k = kk + 1;
                v[k] = cmul<T>(v[k], tw[k * step]);
            });
        }
        dft_any<R, T>(v);
        const int q0 = (jb - jm) * R + jm;
        if constexpr (PHASE == 3) {
            cplx<T>* y = reinterpret_cast<cplx<T>*>(a.out) + (row0 + r) * a.stride_out;
            const T sx = (T)a.scale, sy = a.conj_out ? -sx : sx;
            static_for<R>([&](auto kk) {
                const int q = q0 + kk * Ns;
                if (q < n) {
                    cplx<T> p = v[kk];
                    p.y = -p.y;
                    p = cmul<T>(p, chirp[q]);
                    p.x *= sx;
                    p.y *= sy;
                    y[q] = p;
                }
            });
        } else {
            static_for<R>([&](auto kk) {
                const int q = q0 + kk * Ns;
                cplx<T> p = v[kk];
                if constexpr (PHASE == 2) {
                    p = cmul<T>(p, reinterpret_cast<const cplx<T>*>(a.bhat)[q]);
                    p.y = -p.y;
                }
                dst[r * m + q] = p;
            });
        }
    }
}

template <typename T, int NT, int PHASE, typename... Args>
__device__ __forceinline__ void blue_switch(int R, Args&&... args) {
    switch (R) {
        case 2: blue_stage<2, T, NT, PHASE>(args...); break;
        case 3: blue_stage<3, T, NT, PHASE>(args...); break;
        case 4: blue_stage<4, T, NT, PHASE>(args...); break;
        case 5: blue_stage<5, T, NT, PHASE>(args...); break;
        case 6: blue_stage<6, T, NT, PHASE>(args...); break;
        case 7: blue_stage<7, T, NT, PHASE>(args...); break;
        case 8: blue_stage<8, T, NT, PHASE>(args...); break;
        case 9: blue_stage<9, T, NT, PHASE>(args...); break;
        case 10: blue_stage<10, T, NT, PHASE>(args...); break;
        case 12: blue_stage<12, T, NT, PHASE>(args...); break;
        case 14: blue_stage<14, T, NT, PHASE>(args...); break;
        case 15: blue_stage<15, T, NT, PHASE>(args...); break;
        default: blue_stage<16, T, NT, PHASE>(args...); break;
    }
}

template <typename T, int NT, int OCC>
__global__ void __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(OCC))) fft_bluestein_kernel(const BlueArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cplx<T>* lds = reinterpret_cast<cplx<T>*>(smem);
    const int W = a.W, half = W * a.m, ns = a.nstages;
    const long long row0 = (long long)blockIdx.x * W;
    const int nrows = (int)((a.rows - row0) < W ? (a.rows - row0) : W);
    int cur = 0;
    for (int pass = 0; pass < 2; ++pass) {
        int Ns = 1;
        for (int s = 0; s < ns; ++s) {
            const int R = a.radix[s];
            const float ipr = a.inv_per_row[s], ins = a.inv_ns[s];
            const cplx<T>* src = lds + cur * half;
            cplx<T>* dst = lds + (cur ^ 1) * half;
            const bool first = pass == 0 && s == 0, turn = pass == 0 && s == ns - 1, last = pass == 1 && s == ns - 1;
            if (first) blue_switch<T, NT, 0>(R, a, row0, nrows, src, dst, Ns, ipr, ins);
            else if (turn) blue_switch<T, NT, 2>(R, a, row0, nrows, src, dst, Ns, ipr, ins);
            else if (last) blue_switch<T, NT, 3>(R, a, row0, nrows, src, dst, Ns, ipr, ins);
            else blue_switch<T, NT, 1>(R, a, row0, nrows, src, dst, Ns, ipr, ins);
            if (!last) __syncthreads();
            Ns *= R;
            cur ^= 1;
        }
    }
}

}  // namespace

// padded length for rows of n points: the smooth m in [2 n - 1, tile] with the least m * stages, a power of two at a 12 % discount (at least two stages: the first
// and the last stage of a transform are different code); 0 if none
// Round 4: the padded rows may take the whole LDS of a CU (two buffers of up to 10000 fp32 / 5000 fp64 points = 160 KB, one work-group
// of 1024 threads per CU), which brings every n <= 5000 (fp32) / 2500 (fp64) into the one-launch form: n = 4099 ran the five-launch
// composition at 0.029 of the roofline before (the reference's TODO.txt:8).
constexpr int kBluePoints32 = 10000, kBluePoints64 = 5000;

extern "C" int mifft_bluestein_padded_impl(int f64, int n) {
    const int full = f64 ? kBluePoints64 : kBluePoints32;
    if (n < 2 || 2 * (long long)n - 1 > full) return 0;
    int radix[kMaxStages], best = 0;
    double best_cost = 1e30;
    const int lo = 2 * n - 1 < 4 ? 4 : 2 * n - 1;
    for (int m = lo; m <= full; ++m) {
        const int ns = factor(m, radix);
        if (ns < 2) continue;
        // (a power of two's radix-16 / 8 stages are cheaper per point than radix 15 / 14 / 9 ones: n = 1009 measured 9.3 % of the
        // roofline padded to 2048 against 8.2 % padded to 2025; n = 513: 5.4 % padded to 2048, 8.2 % padded to 1050)
        const double cost = (double)m * ns * ((m & (m - 1)) ? 1.0 : 0.88);
        if (cost < best_cost) { best_cost = cost; best = m; }
        if ((m & (m - 1)) == 0) break;           // nothing beyond the first power of two can be cheaper
    }
    return best;
}

// 0 if m is a padded length the one-launch Bluestein kernel takes (smooth, at least two stages, two buffers of m points fit the LDS)
extern "C" int mifft_bluestein_len_supported_impl(int f64, int m) {
    int radix[kMaxStages];
    if (m < 4 || m > (f64 ? kBluePoints64 : kBluePoints32)) return -2;
    return factor(m, radix) >= 2 ? 0 : -2;
}

extern "C" int mifft_bluestein_launch(int f64, int n, int m, long long rows, long long stride_in, long long stride_out, const void* in,
                                      void* out, const void* tw, const void* chirp, const void* bhat, int flags, double scale, hipStream_t s) {
    BlueArgs a;
    a.nstages = factor(m, a.radix);
    if (a.nstages < 2 || m < 2 * n - 1) return -2;
    a.in = in; a.out = out; a.tw = tw; a.chirp = chirp; a.bhat = bhat;
    a.rows = rows; a.stride_in = stride_in; a.stride_out = stride_out;
    a.n = n; a.m = m; a.scale = scale;
    a.conj_in = flags & 1; a.conj_out = (flags >> 1) & 1;
    for (int i = 0, nsx = 1; i < a.nstages; ++i) {
        a.inv_per_row[i] = 1.0f / (float)(m / a.radix[i]);
        a.inv_ns[i] = 1.0f / (float)nsx;
        nsx *= a.radix[i];
    }
    const int full = f64 ? kTilePoints64 : kTilePoints32;
    if (m > (f64 ? kBluePoints64 : kBluePoints32)) return -2;
    const bool big = m > full;                     // beyond the 64 KiB tiles: one row per work-group, 1024 threads, up to 160 KB of LDS
    const int cap = big ? m : full / (m <= full / 2 ? 2 : 1);
    int W = cap / m;
    if (W < 1) W = 1;
    if (W > rows) W = (int)rows;
    a.W = W;
    const long long blocks = (rows + W - 1) / W;
    if (blocks <= 0) return 0;
    if (blocks > 2147483647ll) return -1;
    const size_t lds_bytes = 2 * (size_t)W * m * (f64 ? 16 : 8);
    if (big) {
        // dynamic LDS beyond 64 KiB has to be asked for once per kernel (per device)
        static thread_local int granted[2][16] = {{0}};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return (int)hipGetLastError();
        if (dev < 0 || dev >= 16 || !granted[f64][dev]) {     // (devices beyond the table: asked for at every launch)
            const void* fn = f64 ? reinterpret_cast<const void*>(&fft_bluestein_kernel<double, 1024, 1>)
                                 : reinterpret_cast<const void*>(&fft_bluestein_kernel<float, 1024, 1>);
            const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return (int)e;
            if (dev >= 0 && dev < 16) granted[f64][dev] = 1;
        }
        if (f64) hipLaunchKernelGGL((fft_bluestein_kernel<double, 1024, 1>), dim3((unsigned)blocks), dim3(1024), lds_bytes, s, a);
        else hipLaunchKernelGGL((fft_bluestein_kernel<float, 1024, 1>), dim3((unsigned)blocks), dim3(1024), lds_bytes, s, a);
        return (int)hipGetLastError();
    }
    if (f64) hipLaunchKernelGGL((fft_bluestein_kernel<double, kThreads, OCC64>), dim3((unsigned)blocks), dim3(kThreads), lds_bytes, s, a);
    else hipLaunchKernelGGL((fft_bluestein_kernel<float, kThreads, OCC32>), dim3((unsigned)blocks), dim3(kThreads), lds_bytes, s, a);
    return (int)hipGetLastError();
}

// flags: bit 0 conjugate on load, bit 1 conjugate on store (inverse transform = 3).  inner > 1: lines of a strided axis.
extern "C" int mifft_mixed_launch(int f64, int n, long long rows, long long stride_in, long long stride_out, long long inner,
                                  const void* in, void* out, const void* tw, int flags, double scale, hipStream_t s) {
    return mixed_launch_impl(f64, n, rows, stride_in, stride_out, inner, in, out, tw, flags, scale, inner > 1 ? 1 : 0, nullptr, nullptr, 0, s);
}

// LONG smooth transforms, N = n1 * n2 beyond one tile, as two launches (four-step form, no separate transposition):
//     x[i1 * n2 + i2]  --lines of n1 points, n2 apart; stored as ROWS, times w(N)^(k1 * i2)-->  t[i2 * n1 + k1]
//                      --lines of n2 points, n1 apart, in place order-->                        X[k1 + n1 * k2]
// The split: both factors fit a tile with as many adjacent lines as possible (the lines of a tile are what makes either pass
// coalesced); 0 and the factors if both passes get >= 8 lines (fp64: 4) per tile, else unsupported.
extern "C" int mifft_mixed_long_split_impl(int f64, long long n, int* n1, int* n2) {
    const int full = f64 ? kTilePoints64 : kTilePoints32;
    if (n < 4 || n > (1ll << 24)) return -2;
    long long m = n;
    for (int c : {2, 3, 5, 7}) while (m % c == 0) m /= c;
    if (m != 1) return -2;
    int best = 0, b1 = 0, b2 = 0;
    for (long long d = 2; d <= full; ++d) {
        if (n % d) continue;
        const long long e = n / d;
        if (e < 2 || e > full) continue;
        const int wa = lines_per_tile(f64, (int)d, e, true), wb = lines_per_tile(f64, (int)e, d, false);
        const int score = wa < wb ? wa : wb;
        if (score > best) { best = score; b1 = (int)d; b2 = (int)e; }
    }
    if (best < (f64 ? 4 : 8)) return -2;
    *n1 = b1; *n2 = b2;
    return 0;
}

// `mid`: where the first pass leaves the transposed array -- `out` itself for an out-of-place transform (the second pass then runs in
// place), a scratch array of the same size for an in-place one.
extern "C" int mifft_mixed_long_launch(int f64, int n1, int n2, long long batch, const void* in, void* mid, void* out, const void* tw1,
                                       const void* tw2, const void* tw_lo, const void* tw_hi, int tw_shift, int flags, double scale,
                                       hipStream_t s) {
    int rc = mixed_launch_impl(f64, n1, batch * n2, n1, n1, n2, in, mid, tw1, flags & 1, 1.0, 2, tw_lo, tw_hi, tw_shift, s);
    if (rc) return rc;
    return mixed_launch_impl(f64, n2, batch * n1, n2, n2, n1, mid, out, tw2, flags & 2, scale, 1, nullptr, nullptr, 0, s);
}

// Mixed-radix rows: contiguous transforms of SMOOTH length n = 2^a 3^b 5^c 7^d (the reference's TODO.txt:8, "support for
// non-power-of-2 sized arrays"), one LDS-resident Stockham transform per row with radix-3 / 5 / 7 butterflies next to the
// power-of-two ones -- instead of Bluestein's three padded power-of-two transforms and four streaming copies
// (pyfft_amd/generic.py), which stays for lengths with a larger prime factor.
//
// A work-group owns W consecutive rows of n points in LDS (W * n <= 4096 fp32 / 2048 fp64 points, two buffers of that size: 64 KiB,
// two work-groups per CU); every stage reads its butterflies' R operands from one buffer and writes the other at the autosort position:
//     stage with radix R, Ns = product of the earlier radices:  butterfly jb < n / R
//         reads   x[jb + k * n / R]                                   k < R
//         twiddle w(n)^(k * (jb mod Ns) * n / (Ns * R))               (table of n entries, float64-evaluated on the host)
//         writes  y[(jb div Ns) * Ns * R + (jb mod Ns) + k * Ns]
// (the same algebra as fft_tile.hpp, without the power-of-two shortcuts: lengths are run-time values, the radix list comes with
// the launch).  The inverse is conj -> forward -> conj.  Interleaved data, in place or out of place, rows `stride` apart.
#include <hip/hip_runtime.h>
#include "../../include/mifft.h"
#include "fft_butterfly.hpp"

namespace {
using namespace mifft;

constexpr int kMaxStages = 12;

struct MixedArgs {
    const void* in;
    void* out;
    const void* tw;        // n entries w(n)^m
    long long rows, stride_in, stride_out;
    long long inner;       // 1: contiguous rows `stride` apart.  > 1: LINES of a strided axis -- line L = o * inner + j starts at element
                           // o * n * inner + j and its points are `inner` elements apart (stride_in / stride_out unused)
    int n, W, nstages, inverse;
    int conj_in, conj_out; // conjugate on load / on store (inverse = both; an N-D plan conjugates once at either end)
    int radix[kMaxStages];
    float inv_n;           // 1 / n, 1 / (n / R) and 1 / Ns per stage: index divisions as one float multiply (indices < 2^22, exact
    float inv_per_row[kMaxStages], inv_ns[kMaxStages];   // with the + 0.5 below)
    double scale;
};

// q = a / d for 0 <= a < 2^22, inv = 1.0f / d
__device__ __forceinline__ int fast_div(int a, float inv) { return (int)(((float)a + 0.5f) * inv); }

// odd radices: forward DFT of R points in registers (e = exp(-2 pi i / R) powers as literals)
template <typename T> __device__ __forceinline__ void dft3(cplx<T>* v) {
    const T c = (T)-0.5, s = (T)0.86602540378443864676;
    const cplx<T> t = v[1] + v[2], d = v[1] - v[2];
    const cplx<T> m = {v[0].x + c * t.x, v[0].y + c * t.y};
    v[0] = v[0] + t;
    v[1] = cplx<T>{m.x + s * d.y, m.y - s * d.x};     // m - i s d
    v[2] = cplx<T>{m.x - s * d.y, m.y + s * d.x};
}
template <typename T> __device__ __forceinline__ void dft5(cplx<T>* v) {
    const T c1 = (T)0.30901699437494742410, c2 = (T)-0.80901699437494742410, s1 = (T)0.95105651629515357212, s2 = (T)0.58778525229247312917;
    const cplx<T> a1 = v[1] + v[4], b1 = v[1] - v[4], a2 = v[2] + v[3], b2 = v[2] - v[3];
    const cplx<T> m1 = {v[0].x + c1 * a1.x + c2 * a2.x, v[0].y + c1 * a1.y + c2 * a2.y};
    const cplx<T> m2 = {v[0].x + c2 * a1.x + c1 * a2.x, v[0].y + c2 * a1.y + c1 * a2.y};
    const cplx<T> n1 = {s1 * b1.x + s2 * b2.x, s1 * b1.y + s2 * b2.y};
    const cplx<T> n2 = {s2 * b1.x - s1 * b2.x, s2 * b1.y - s1 * b2.y};
    v[0] = v[0] + a1 + a2;
    v[1] = cplx<T>{m1.x + n1.y, m1.y - n1.x};          // m1 - i n1
    v[4] = cplx<T>{m1.x - n1.y, m1.y + n1.x};
    v[2] = cplx<T>{m2.x + n2.y, m2.y - n2.x};
    v[3] = cplx<T>{m2.x - n2.y, m2.y + n2.x};
}
template <typename T> __device__ __forceinline__ void dft7(cplx<T>* v) {
    const T c1 = (T)0.62348980185873353053, c2 = (T)-0.22252093395631440429, c3 = (T)-0.90096886790241912624;
    const T s1 = (T)0.78183148246802980871, s2 = (T)0.97492791218182360702, s3 = (T)0.43388373911755812048;
    const cplx<T> a1 = v[1] + v[6], b1 = v[1] - v[6], a2 = v[2] + v[5], b2 = v[2] - v[5], a3 = v[3] + v[4], b3 = v[3] - v[4];
    auto re = [&](T x1, T x2, T x3) { return cplx<T>{v[0].x + x1 * a1.x + x2 * a2.x + x3 * a3.x, v[0].y + x1 * a1.y + x2 * a2.y + x3 * a3.y}; };
    auto im = [&](T y1, T y2, T y3) { return cplx<T>{y1 * b1.x + y2 * b2.x + y3 * b3.x, y1 * b1.y + y2 * b2.y + y3 * b3.y}; };
    const cplx<T> m1 = re(c1, c2, c3), m2 = re(c2, c3, c1), m3 = re(c3, c1, c2);
    const cplx<T> n1 = im(s1, s2, s3), n2 = im(s2, -s3, -s1), n3 = im(s3, -s1, s2);
    v[0] = v[0] + a1 + a2 + a3;
    v[1] = cplx<T>{m1.x + n1.y, m1.y - n1.x};
    v[6] = cplx<T>{m1.x - n1.y, m1.y + n1.x};
    v[2] = cplx<T>{m2.x + n2.y, m2.y - n2.x};
    v[5] = cplx<T>{m2.x - n2.y, m2.y + n2.x};
    v[3] = cplx<T>{m3.x + n3.y, m3.y - n3.x};
    v[4] = cplx<T>{m3.x - n3.y, m3.y + n3.x};
}
template <int R, typename T> __device__ __forceinline__ void dft_any(cplx<T>* v) {
    if constexpr (R == 3) dft3<T>(v);
    else if constexpr (R == 5) dft5<T>(v);
    else if constexpr (R == 7) dft7<T>(v);
    else Dft<R, T>::run(v);
}

// one butterfly of radix R, row-local index jb: operands from `src` (twiddled), DFT in registers, results to `dst` at the autosort
// position.  GIN / GOUT: `src` / `dst` is the row in GLOBAL memory (first / last stage of the row form: consecutive butterflies read
// consecutive points of every operand, and the last stage, Ns = n / R, writes consecutive points of every result), with the
// conjugation of the inverse direction / the scale folded in.
template <int R, typename T, bool GIN, bool GOUT>
__device__ __forceinline__ void stage_butterfly(const cplx<T>* src, cplx<T>* dst, const cplx<T>* tw, int LR, int Ns, float inv_ns, int jb,
                                                T csign, T sx, T sy) {
    const int jm = jb - fast_div(jb, inv_ns) * Ns;
    cplx<T> v[R];
    static_for<R>([&](auto kk) { v[kk] = src[jb + kk * LR]; });
    if constexpr (GIN) static_for<R>([&](auto kk) { v[kk].y *= csign; });
    if (Ns > 1) {
        const int step = jm * (LR / Ns);                 // jm * n / (Ns * R)
        static_for<R - 1>([&](auto kk) {
            constexpr int k = kk + 1;
            v[k] = cmul<T>(v[k], tw[k * step]);
        });
    }
    dft_any<R, T>(v);
    cplx<T>* q = dst + (jb - jm) * R + jm;
    static_for<R>([&](auto kk) {
        cplx<T> p = v[kk];
        if constexpr (GOUT) {
            p.x *= sx;
            p.y *= sy;
        }
        q[kk * Ns] = p;
    });
}

template <typename T, bool GIN, bool GOUT>
__device__ __forceinline__ void stage_switch(int R, const cplx<T>* sr, cplx<T>* dr, const cplx<T>* tw, int per_row, int Ns, float ins, int jb,
                                             T csign, T sx, T sy) {
    switch (R) {
        case 2: stage_butterfly<2, T, GIN, GOUT>(sr, dr, tw, per_row, Ns, ins, jb, csign, sx, sy); break;
        case 3: stage_butterfly<3, T, GIN, GOUT>(sr, dr, tw, per_row, Ns, ins, jb, csign, sx, sy); break;
        case 4: stage_butterfly<4, T, GIN, GOUT>(sr, dr, tw, per_row, Ns, ins, jb, csign, sx, sy); break;
        case 5: stage_butterfly<5, T, GIN, GOUT>(sr, dr, tw, per_row, Ns, ins, jb, csign, sx, sy); break;
        case 7: stage_butterfly<7, T, GIN, GOUT>(sr, dr, tw, per_row, Ns, ins, jb, csign, sx, sy); break;
        case 8: stage_butterfly<8, T, GIN, GOUT>(sr, dr, tw, per_row, Ns, ins, jb, csign, sx, sy); break;
        default: stage_butterfly<16, T, GIN, GOUT>(sr, dr, tw, per_row, Ns, ins, jb, csign, sx, sy); break;
    }
}

// Two LDS buffers of W * n points: a stage reads one and writes the other (one barrier per stage).
template <typename T, int NT> __global__ void __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(4))) fft_mixed_kernel(const MixedArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cplx<T>* lds = reinterpret_cast<cplx<T>*>(smem);
    const cplx<T>* tw = reinterpret_cast<const cplx<T>*>(a.tw);
    const int n = a.n, W = a.W, tid = threadIdx.x;
    const int half = W * n;
    const long long row0 = (long long)blockIdx.x * W;
    const int nrows = (int)((a.rows - row0) < W ? (a.rows - row0) : W);
    const T csign = a.conj_in ? (T)-1 : (T)1;
    const long long inner = a.inner;
    long long line_base = 0;      // strided axis: the tile's first line (its W lines are adjacent in memory: coalesced across lines)
    float inv_rows = 1.0f;
    if (inner > 1) {
        // (the launcher makes W divide `inner`, so a tile never straddles two values of o)
        const long long o = row0 / inner, j0 = row0 - o * inner;
        line_base = o * (long long)n * inner + j0;
        inv_rows = 1.0f / (float)nrows;
    }
    const T sx = (T)a.scale, sy = a.conj_out ? -sx : sx;
    // register edges (the first stage reads HBM, the last one writes HBM) for fp64 rows: N = 1000 31.1 -> 34.4 %; fp32 rows measured
    // better through the linear 8-byte staging copy (30.2 against 28.6 %) and keep it
    const bool rowform = inner == 1 && sizeof(T) == 8;
    if (inner == 1 && !rowform) {
        // rows -> LDS (consecutive threads, consecutive points)
        for (int e = tid; e < nrows * n; e += NT) {
            const int r = fast_div(e, a.inv_n), i = e - r * n;
            cplx<T> p = reinterpret_cast<const cplx<T>*>(a.in)[(row0 + r) * a.stride_in + i];
            p.y *= csign;
            lds[e] = p;
        }
        __syncthreads();
    } else if (!rowform) {
        // lines -> LDS (consecutive threads, consecutive LINES of the same point index)
        for (int e = tid; e < nrows * n; e += NT) {
            const int i = fast_div(e, inv_rows), c = e - i * nrows;
            cplx<T> p = reinterpret_cast<const cplx<T>*>(a.in)[line_base + (long long)i * inner + c];
            p.y *= csign;
            lds[c * n + i] = p;
        }
        __syncthreads();
    }
    int Ns = 1, cur = 0;
    for (int s = 0; s < a.nstages; ++s) {
        const int R = a.radix[s];
        const int per_row = n / R, total = nrows * per_row;
        const float ipr = a.inv_per_row[s], ins = a.inv_ns[s];
        const cplx<T>* src = lds + cur * half;
        cplx<T>* dst = lds + (cur ^ 1) * half;
        const bool gin = rowform && s == 0, gout = rowform && s == a.nstages - 1;
        for (int j = tid; j < total; j += NT) {
            const int r = fast_div(j, ipr), jb = j - r * per_row;
            const cplx<T>* sr = gin ? reinterpret_cast<const cplx<T>*>(a.in) + (row0 + r) * a.stride_in : src + r * n;
            cplx<T>* dr = gout ? reinterpret_cast<cplx<T>*>(a.out) + (row0 + r) * a.stride_out : dst + r * n;
            if (gin && gout) stage_switch<T, true, true>(R, sr, dr, tw, per_row, Ns, ins, jb, csign, sx, sy);
            else if (gin) stage_switch<T, true, false>(R, sr, dr, tw, per_row, Ns, ins, jb, csign, sx, sy);
            else if (gout) stage_switch<T, false, true>(R, sr, dr, tw, per_row, Ns, ins, jb, csign, sx, sy);
            else stage_switch<T, false, false>(R, sr, dr, tw, per_row, Ns, ins, jb, csign, sx, sy);
        }
        __syncthreads();
        Ns *= R;
        cur ^= 1;
    }
    if (rowform) return;
    const cplx<T>* res = lds + cur * half;
    if (inner == 1) {
        for (int e = tid; e < nrows * n; e += NT) {
            const int r = fast_div(e, a.inv_n), i = e - r * n;
            cplx<T> p = res[e];
            p.x *= sx;
            p.y *= sy;
            reinterpret_cast<cplx<T>*>(a.out)[(row0 + r) * a.stride_out + i] = p;
        }
    } else {
        for (int e = tid; e < nrows * n; e += NT) {
            const int i = fast_div(e, inv_rows), c = e - i * nrows;
            cplx<T> p = res[c * n + i];
            p.x *= sx;
            p.y *= sy;
            reinterpret_cast<cplx<T>*>(a.out)[line_base + (long long)i * inner + c] = p;
        }
    }
}

// radix list of n; 0 if n has a prime factor beyond 7.  Odd radices FIRST, the powers of two last (largest last): the first
// stage writes with stride R -- 10 / 14 dwords for radix 5 / 7 spread over the LDS banks, 16 or 32 dwords for radix 8 / 16 would
// put a wave on 4 or 2 of the 64 banks -- and the last stage (Ns = n / R) writes consecutive addresses whatever its radix.
int factor(int n, int* radix) {
    int ns = 0;
    for (int c : {7, 5, 3}) {
        while (n % c == 0) {
            if (ns >= kMaxStages) return 0;
            radix[ns++] = c;
            n /= c;
        }
    }
    if (n & (n - 1)) return 0;             // what is left must be a power of two
    int tail[kMaxStages], nt = 0;
    while (n > 1) {
        const int r = n % 16 == 0 ? 16 : n % 8 == 0 ? 8 : n % 4 == 0 ? 4 : 2;
        if (ns + nt >= kMaxStages) return 0;
        tail[nt++] = r;
        n /= r;
    }
    for (int i = nt - 1; i >= 0; --i) radix[ns++] = tail[i];   // smallest power of two first, the largest last
    return ns;
}

constexpr int kTilePoints32 = 4096, kTilePoints64 = 2048, kThreads = 256;

}  // namespace

// 0 if rows of n points have a mixed-radix kernel: n = 2^a 3^b 5^c 7^d, 2 <= n <= 4096 (fp32) / 2048 (fp64)
extern "C" int mifft_mixed_supported_impl(int f64, int n) {
    int radix[kMaxStages];
    if (n < 2 || n > (f64 ? kTilePoints64 : kTilePoints32)) return -2;
    return factor(n, radix) ? 0 : -2;
}

// flags: bit 0 conjugate on load, bit 1 conjugate on store (inverse transform = 3).  inner > 1: lines of a strided axis.
extern "C" int mifft_mixed_launch(int f64, int n, long long rows, long long stride_in, long long stride_out, long long inner,
                                  const void* in, void* out, const void* tw, int flags, double scale, hipStream_t s) {
    MixedArgs a;
    a.nstages = factor(n, a.radix);
    if (!a.nstages) return -2;
    a.in = in; a.out = out; a.tw = tw;
    a.rows = rows; a.stride_in = stride_in; a.stride_out = stride_out;
    a.n = n; a.inverse = (flags & 3) == 3; a.scale = scale;
    a.conj_in = flags & 1; a.conj_out = (flags >> 1) & 1;
    a.inner = inner < 1 ? 1 : inner;
    a.inv_n = 1.0f / (float)n;
    for (int i = 0, nsx = 1; i < a.nstages; ++i) {
        a.inv_per_row[i] = 1.0f / (float)(n / a.radix[i]);
        a.inv_ns[i] = 1.0f / (float)nsx;
        nsx *= a.radix[i];
    }
    // tiles of half the capacity (32 KiB of LDS, four work-groups per CU) whenever a row fits one: the kernel is latency-bound
    // (two work-groups per CU: N = 1000 fp32 18.5 % of the roofline, four: 30.2 %)
    const int full = f64 ? kTilePoints64 : kTilePoints32;
    const int cap = full / (n <= full / 2 ? 2 : 1);          // (quarter tiles, eight work-groups per CU: 29.0 against 30.2 %)
    int W = cap / n;
    if (W < 1) W = 1;
    if (W > rows) W = (int)rows;
    if (a.inner > 1) {
        // a tile's W lines are adjacent and share one o: the largest W <= cap / n that divides `inner`
        while (W > 1 && a.inner % W) --W;
    }
    a.W = W;
    const long long blocks = (rows + W - 1) / W;
    if (blocks <= 0) return 0;
    if (blocks > 2147483647ll) return -1;
    const size_t lds_bytes = 2 * (size_t)W * n * (f64 ? 16 : 8);
    if (f64) hipLaunchKernelGGL((fft_mixed_kernel<double, kThreads>), dim3((unsigned)blocks), dim3(kThreads), lds_bytes, s, a);
    else hipLaunchKernelGGL((fft_mixed_kernel<float, kThreads>), dim3((unsigned)blocks), dim3(kThreads), lds_bytes, s, a);
    return (int)hipGetLastError();
}

// Whole SMOOTH 2-D / 3-D transforms in ONE launch (round 4; the reference's TODO.txt:8 "non-power-of-2 sized arrays" for N-D shapes):
// every axis length of the form 2^a 3^b 5^c 7^d (or 1), the whole transform -- or several of them -- in one work-group's LDS:
// (100, 100) is 80 KB, one tile.  Round 3 ran one launch PER AXIS on the user's buffers (mifft_launch_mixed_lines: every launch one
// HBM round trip, so two / three launches bound such shapes at a half / a third of the row figure: (100, 100) 0.286).
//
// A tile = W consecutive transforms of P = nx * ny * nz points, dense in HBM and in LDS alike ([w][z][y][x], x fastest).  The stage
// loop of fft_mixed.hip runs over the stages of the x axis, then y, then z; a stage of the axis with length n and `inner` =
// product of the faster axes works on all W * P / n lines of that axis in the tile:
//     butterfly j -> (jin, jb, o):  jin = j mod inner,  jb = (j div inner) mod (n / R),  o = j div (inner * n / R)
//     line base = o * n * inner + jin,  point i of the line at base + i * inner
// so that consecutive lanes take consecutive butterflies of a row (inner == 1: the x axis) or adjacent lines (inner > 1):
// both the HBM side of the first / last stage and the LDS side of every stage are walked with unit stride across the wave.
// The first stage of all reads HBM (conjugated for the inverse), the last stage of all writes it (scaled, conjugated).
//
// ONE LDS buffer: in a stage every thread first reads the operands of ALL its butterflies into registers (up to 32 fp32 / 16 fp64
// points per thread: 32 / R butterflies of radix R), a barrier, then computes and writes the results to their autosort positions
// in the same buffer, a barrier.  (First form of the round: two ping-pong buffers, one barrier per stage -- a (100, 100) tile then
// fills the LDS of a CU, one work-group of 1024 threads per CU with nothing to overlap its load, stages and store: 0.303 of the
// roofline, as slow as the two launches it replaced; profiles/r04_d_mixed_radix.log.)
#include "fft_mixed.hpp"

namespace {

constexpr int kNdMaxStages = 3 * 6;
// points a thread holds across the barrier of a stage
template <typename T> struct NdRegs { static constexpr int value = sizeof(T) == 4 ? 32 : 16; };
// the largest tile: every point of it in the registers of 512 threads, and in LDS once
constexpr int kNdTilePoints32 = 512 * 32, kNdTilePoints64 = 512 * 16;

struct MixedNdArgs {
    const void* in;
    void* out;
    const void* tw[3];          // w(len)^m of the x / y / z axis (null for an axis of length 1)
    long long transforms;       // batch
    int P, W, nstages;
    int conj_in, conj_out;
    unsigned char st_axis[kNdMaxStages];
    short st_radix[kNdMaxStages];
    int st_n[kNdMaxStages], st_inner[kNdMaxStages], st_ns[kNdMaxStages];
    float inv_inner[kNdMaxStages], inv_lr[kNdMaxStages], inv_ns[kNdMaxStages];
    double scale;
};

template <int R, typename T, int NT, bool GIN, bool GOUT>
__device__ __forceinline__ void nd_stage(const cplx<T>* src, cplx<T>* dst, const cplx<T>* tw, const int n, const int inner, const int Ns,
                                         const float inv_inner, const float inv_lr, const float inv_ns, const int total, const T csign,
                                         const T sx, const T sy) {
    constexpr int B = NdRegs<T>::value / R > 0 ? NdRegs<T>::value / R : 1;     // butterflies per thread (the launcher sizes the tile for it)
    const int LR = n / R;
    cplx<T> v[B][R];
    int base[B], jbs[B];
    static_for<B>([&](auto bb) {
        constexpr int b = bb;
        const int j = threadIdx.x + b * NT;
        int jin = 0, t = j;
        if (inner > 1) {
            t = fast_div(j, inv_inner);
            jin = j - t * inner;
        }
        const int o = fast_div(t, inv_lr), jb = t - o * LR;
        base[b] = o * n * inner + jin;
        jbs[b] = jb;
        if (j < total) static_for<R>([&](auto kk) { v[b][kk] = src[base[b] + (jb + kk * LR) * inner]; });
    });
    if constexpr (!GIN) __syncthreads();          // every operand of the stage is in registers: the buffer may be overwritten
    static_for<B>([&](auto bb) {
        constexpr int b = bb;
        const int j = threadIdx.x + b * NT;
        if (j < total) {
            const int jb = jbs[b];
            const int jm = jb - fast_div(jb, inv_ns) * Ns;
            if constexpr (GIN) static_for<R>([&](auto kk) { v[b][kk].y *= csign; });
            if (Ns > 1) {
                const int step = jm * (LR / Ns);
                static_for<R - 1>([&](auto kk) {
                    constexpr int k = kk + 1;
                    v[b][k] = cmul<T>(v[b][k], tw[k * step]);
                });
            }
            dft_any<R, T>(v[b]);
            const int q0 = (jb - jm) * R + jm;
            static_for<R>([&](auto kk) {
                cplx<T> p = v[b][kk];
                if constexpr (GOUT) {
                    p.x *= sx;
                    p.y *= sy;
                }
                dst[base[b] + (q0 + kk * Ns) * inner] = p;
            });
        }
    });
    if constexpr (!GOUT) __syncthreads();         // the next stage reads what this one wrote
}

template <typename T, int NT, bool GIN, bool GOUT, typename... Args>
__device__ __forceinline__ void nd_switch(int R, Args&&... args) {
    switch (R) {
        case 2: nd_stage<2, T, NT, GIN, GOUT>(args...); break;
        case 3: nd_stage<3, T, NT, GIN, GOUT>(args...); break;
        case 4: nd_stage<4, T, NT, GIN, GOUT>(args...); break;
        case 5: nd_stage<5, T, NT, GIN, GOUT>(args...); break;
        case 6: nd_stage<6, T, NT, GIN, GOUT>(args...); break;
        case 7: nd_stage<7, T, NT, GIN, GOUT>(args...); break;
        case 8: nd_stage<8, T, NT, GIN, GOUT>(args...); break;
        case 9: nd_stage<9, T, NT, GIN, GOUT>(args...); break;
        case 10: nd_stage<10, T, NT, GIN, GOUT>(args...); break;
        case 12: nd_stage<12, T, NT, GIN, GOUT>(args...); break;
        case 14: nd_stage<14, T, NT, GIN, GOUT>(args...); break;
        case 15: nd_stage<15, T, NT, GIN, GOUT>(args...); break;
        default: nd_stage<16, T, NT, GIN, GOUT>(args...); break;
    }
}

// (four waves per SIMD = 128 VGPRs: two work-groups of 512 threads, or four of 256, per CU; without the bound the kernels take 139)
template <typename T, int NT>
__global__ void __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(4))) fft_mixed_nd_kernel(const MixedNdArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cplx<T>* lds = reinterpret_cast<cplx<T>*>(smem);
    const long long t0 = (long long)blockIdx.x * a.W;
    const int nw = (int)((a.transforms - t0) < a.W ? (a.transforms - t0) : a.W);
    const int points = nw * a.P;
    const cplx<T>* gin = reinterpret_cast<const cplx<T>*>(a.in) + t0 * a.P;
    cplx<T>* gout = reinterpret_cast<cplx<T>*>(a.out) + t0 * a.P;
    const T csign = a.conj_in ? (T)-1 : (T)1;
    const T sx = (T)a.scale, sy = a.conj_out ? -sx : sx;
    for (int s = 0; s < a.nstages; ++s) {
        const int R = a.st_radix[s];
        const cplx<T>* tw = reinterpret_cast<const cplx<T>*>(a.tw[a.st_axis[s]]);
        const bool first = s == 0, last = s == a.nstages - 1;
        const cplx<T>* src = first ? gin : lds;
        cplx<T>* dst = last ? gout : lds;
        const int total = points / R;
        // (a list has at least two stages -- the launcher sees to it: rows of a single radix, n <= 16, run the row kernel of fft_mixed.hip;
        // the HBM -> HBM form of every radix would be a quarter of this unit's compile time)
        if (first) nd_switch<T, NT, true, false>(R, src, dst, tw, a.st_n[s], a.st_inner[s], a.st_ns[s], a.inv_inner[s], a.inv_lr[s], a.inv_ns[s], total, csign, sx, sy);
        else if (last) nd_switch<T, NT, false, true>(R, src, dst, tw, a.st_n[s], a.st_inner[s], a.st_ns[s], a.inv_inner[s], a.inv_lr[s], a.inv_ns[s], total, csign, sx, sy);
        else nd_switch<T, NT, false, false>(R, src, dst, tw, a.st_n[s], a.st_inner[s], a.st_ns[s], a.inv_inner[s], a.inv_lr[s], a.inv_ns[s], total, csign, sx, sy);
    }
}

// stage list of a shape: the stages of x, then y, then z; 0 if an axis is not smooth or the list is too long.
// *per_thread = the points one thread can hold in EVERY stage of the list: a stage of radix R keeps floor(regs / R) butterflies per
// thread, so a tile on NT threads may have at most NT * min_R (floor(regs / R) * R) points (radix 10 in fp64: 10 of the 16).
int nd_stages(int f64, int nx, int ny, int nz, MixedNdArgs* a, int* per_thread) {
    const int dims[3] = {nx, ny, nz};
    const int regs = f64 ? NdRegs<double>::value : NdRegs<float>::value;
    int ns = 0, inner = 1, hold = regs;
    for (int ax = 0; ax < 3; ++ax) {
        const int n = dims[ax];
        if (n > 1) {
            int radix[kMaxStages];
            const int k = factor(n, radix);
            if (!k || ns + k > kNdMaxStages) return 0;
            int nsx = 1;
            for (int i = 0; i < k; ++i) {
                if (a) {
                    a->st_axis[ns] = (unsigned char)ax;
                    a->st_radix[ns] = (short)radix[i];
                    a->st_n[ns] = n;
                    a->st_inner[ns] = inner;
                    a->st_ns[ns] = nsx;
                    a->inv_inner[ns] = 1.0f / (float)inner;
                    a->inv_lr[ns] = 1.0f / (float)(n / radix[i]);
                    a->inv_ns[ns] = 1.0f / (float)nsx;
                }
                const int h = (regs / radix[i] > 0 ? regs / radix[i] : 1) * radix[i];
                if (h < hold) hold = h;
                nsx *= radix[i];
                ++ns;
            }
        }
        inner *= n;
    }
    if (per_thread) *per_thread = hold;
    return ns;
}

template <typename T, int NT> int launch_nd_kernel(const MixedNdArgs& a, unsigned blocks, size_t lds_bytes, hipStream_t s) {
    // dynamic LDS beyond 64 KiB has to be asked for once per kernel (per device)
    static thread_local int granted[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return (int)hipGetLastError();
    if (dev < 0 || dev >= 16 || !granted[dev]) {     // (devices beyond the table: asked for at every launch)
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fft_mixed_nd_kernel<T, NT>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        if (dev >= 0 && dev < 16) granted[dev] = 1;
    }
    hipLaunchKernelGGL((fft_mixed_nd_kernel<T, NT>), dim3(blocks), dim3(NT), lds_bytes, s, a);
    return (int)hipGetLastError();
}

}  // namespace

// 0 if the (z, y, x) shape runs as one launch: at least two axes longer than 1, every axis smooth, the transform fits a tile
extern "C" int mifft_mixed_nd_supported_impl(int f64, int nx, int ny, int nz) {
    if (nx < 1 || ny < 1 || nz < 1) return -2;
    if ((nx > 1) + (ny > 1) + (nz > 1) < 2) return -2;
    const long long P = (long long)nx * ny * nz;
    if (P > (f64 ? kNdTilePoints64 : kNdTilePoints32)) return -2;
    int hold = 0;
    if (nd_stages(f64, nx, ny, nz, nullptr, &hold) < 2) return -2;
    return P <= 512ll * hold ? 0 : -2;          // the whole transform in the registers of 512 threads in every stage
}

// 0 if dense ROWS of n points (a 1-D smooth length) fit the tile of this kernel: the same stage loop with one axis
extern "C" int mifft_mixed_nd_rows_ok_impl(int f64, int n) {
    if (n < 2 || n > (f64 ? kNdTilePoints64 : kNdTilePoints32)) return -2;
    int hold = 0;
    if (nd_stages(f64, n, 1, 1, nullptr, &hold) < 2) return -2;
    return n <= 512 * hold ? 0 : -2;
}

// flags: bit 0 conjugate on load, bit 1 conjugate on store (inverse transform = 3)
extern "C" int mifft_mixed_nd_launch(int f64, int nx, int ny, int nz, long long transforms, const void* in, void* out, const void* twx,
                                     const void* twy, const void* twz, int flags, double scale, hipStream_t s) {
    MixedNdArgs a;
    int hold = 0;
    a.nstages = nd_stages(f64, nx, ny, nz, &a, &hold);
    if (a.nstages < 2 || (long long)nx * ny * nz > 512ll * hold) return -2;
    a.in = in; a.out = out;
    a.tw[0] = twx; a.tw[1] = twy; a.tw[2] = twz;
    a.transforms = transforms;
    a.P = nx * ny * nz;
    a.conj_in = flags & 1; a.conj_out = (flags >> 1) & 1;
    a.scale = scale;
    // W whole transforms per tile: small tiles (<= 4096 fp32 / 2048 fp64 points, 32 KiB of LDS, 256 threads: four to five work-groups
    // per CU, as the row kernel) when a transform fits one, else tiles of up to 512 threads x the register budget
    const int full = f64 ? kNdTilePoints64 : kNdTilePoints32;
    const int small = full / 4 < 256 * hold ? full / 4 : 256 * hold;          // (both bounded by what the threads can hold in every stage)
    const bool big = a.P > small;
    int cap = big ? (a.P <= full / 2 ? full / 2 : full) : small;
    if (big && cap > 512 * hold) cap = 512 * hold;
    int W = cap / a.P;
    if (W < 1) W = 1;
    if (W > transforms) W = (int)transforms;
    a.W = W;
    const long long blocks = (transforms + W - 1) / W;
    if (blocks <= 0) return 0;
    if (blocks > 2147483647ll) return -1;
    const size_t lds_bytes = (size_t)W * a.P * (f64 ? 16 : 8);
    if (f64) return big ? launch_nd_kernel<double, 512>(a, (unsigned)blocks, lds_bytes, s) : launch_nd_kernel<double, 256>(a, (unsigned)blocks, lds_bytes, s);
    return big ? launch_nd_kernel<float, 512>(a, (unsigned)blocks, lds_bytes, s) : launch_nd_kernel<float, 256>(a, (unsigned)blocks, lds_bytes, s);
}

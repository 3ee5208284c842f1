// Auxiliary element-wise kernels behind the extensions the reference lists as TODO (TODO.txt:6-8): a general strided complex
// copy (gathers the tiles of a bigger array / the lines of one axis into dense rows and scatters them back; optional
// zero padding, conjugation, a per-position complex multiplier and a scale: the chirp steps of Bluestein's algorithm) and the
// row-wise spectrum product.  One thread per element, fastest index along the threads; these passes are plain streaming
// copies (16 bytes per element and side) and are not on the power-of-two hot path.
#include <hip/hip_runtime.h>
#include "../../include/mifft.h"
#include "fft_butterfly.hpp"
#include <cstdint>

namespace {

struct CopyArgs {
    long long dims[6], ss[6], ds[6];
    long long total, src_valid0;
    const void *s0, *s1, *mult;
    void *d0, *d1;
    int ndim, src_split, dst_split, conj_in, conj_out;
    double scale;
};

template <typename T> __global__ void __launch_bounds__(256) aux_copy_kernel(const CopyArgs a) {
    using C = mifft::cplx<T>;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < a.total; id += stride) {
        long long rest = id, so = 0, dof = 0, i0 = 0;
        for (int d = 0; d < a.ndim; ++d) {
            const long long i = rest % a.dims[d];
            rest /= a.dims[d];
            if (d == 0) i0 = i;
            so += i * a.ss[d];
            dof += i * a.ds[d];
        }
        C v = {(T)0, (T)0};
        if (i0 < a.src_valid0) {
            if (a.src_split) {
                v.x = reinterpret_cast<const T*>(a.s0)[so];
                v.y = reinterpret_cast<const T*>(a.s1)[so];
            } else {
                v = reinterpret_cast<const C*>(a.s0)[so];
            }
            if (a.conj_in) v.y = -v.y;
            if (a.mult) v = mifft::cmul<T>(v, reinterpret_cast<const C*>(a.mult)[i0]);
        }
        v.x *= (T)a.scale;
        v.y *= (T)a.scale;
        if (a.conj_out) v.y = -v.y;
        if (a.dst_split) {
            reinterpret_cast<T*>(a.d0)[dof] = v.x;
            reinterpret_cast<T*>(a.d1)[dof] = v.y;
        } else {
            reinterpret_cast<C*>(a.d0)[dof] = v;
        }
    }
}

// Dense-run form: both sides contiguous along dims[0] (stride 1), interleaved, no multiplier and no padding -- the tile
// gather / scatter.  One thread moves V consecutive elements (16 bytes for fp32) and the index arithmetic is 32-bit.
template <typename T, int V> __global__ void __launch_bounds__(256) aux_copy_runs_kernel(const CopyArgs a, unsigned d0v, unsigned totalv) {
    using C = mifft::cplx<T>;
    typedef T VT __attribute__((ext_vector_type(2 * V)));
    const unsigned stride = gridDim.x * blockDim.x;
    for (unsigned id = blockIdx.x * blockDim.x + threadIdx.x; id < totalv; id += stride) {
        unsigned rest = id / d0v;
        const unsigned i0 = (id - rest * d0v) * V;
        long long so = i0, dof = i0;
        for (int d = 1; d < a.ndim; ++d) {
            const unsigned dim = (unsigned)a.dims[d];
            const unsigned q = rest / dim, i = rest - q * dim;
            rest = q;
            so += (long long)i * a.ss[d];
            dof += (long long)i * a.ds[d];
        }
        VT v = *reinterpret_cast<const VT*>(reinterpret_cast<const C*>(a.s0) + so);
        const T sc = (T)a.scale;
        for (int k = 0; k < 2 * V; ++k) v[k] *= ((k & 1) && (a.conj_in != a.conj_out)) ? -sc : sc;
        *reinterpret_cast<VT*>(reinterpret_cast<C*>(a.d0) + dof) = v;
    }
}

template <typename T> __global__ void __launch_bounds__(256) aux_mul_rows_kernel(mifft::cplx<T>* a, const mifft::cplx<T>* b, long long total, long long n) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += stride)
        a[id] = mifft::cmul<T>(a[id], b[id % n]);
}

unsigned grid_for(long long total) {
    long long g = (total + 255) / 256;
    if (g > 256 * 32) g = 256 * 32;
    return (unsigned)(g < 1 ? 1 : g);
}

}  // namespace

extern "C" int mifft_aux_copy_launch(const mifft_copy* c, const void* s0, const void* s1, void* d0, void* d1, hipStream_t s) {
    CopyArgs a;
    a.total = 1;
    for (int d = 0; d < 6; ++d) {
        a.dims[d] = d < c->ndim ? c->dims[d] : 1;
        a.ss[d] = d < c->ndim ? c->src_stride[d] : 0;
        a.ds[d] = d < c->ndim ? c->dst_stride[d] : 0;
        a.total *= a.dims[d];
    }
    a.ndim = c->ndim;
    a.src_valid0 = c->src_valid0 > 0 ? c->src_valid0 : a.dims[0];
    a.s0 = s0; a.s1 = s1; a.d0 = d0; a.d1 = d1;
    a.mult = c->mult;
    a.src_split = c->src_split; a.dst_split = c->dst_split;
    a.conj_in = c->conj_in; a.conj_out = c->conj_out;
    a.scale = c->scale;
    if (a.total <= 0) return 0;
    // dense runs along dims[0] on both sides: the vector form (every offset then is a multiple of V elements)
    const int V = c->precision == MIFFT_F64 ? 1 : 2;
    bool runs = !c->src_split && !c->dst_split && !c->mult && a.src_valid0 >= a.dims[0] && a.ss[0] == 1 && a.ds[0] == 1 &&
                a.dims[0] % V == 0 && a.total / V < 0x7fffffffll &&
                (((uintptr_t)s0 | (uintptr_t)d0) & 15) == 0;
    for (int d = 1; d < a.ndim && runs; ++d) runs = a.ss[d] % V == 0 && a.ds[d] % V == 0 && a.dims[d] < 0x7fffffffll;
    if (runs) {
        const unsigned d0v = (unsigned)(a.dims[0] / V), totalv = (unsigned)(a.total / V);
        if (c->precision == MIFFT_F64) hipLaunchKernelGGL((aux_copy_runs_kernel<double, 1>), dim3(grid_for(totalv)), dim3(256), 0, s, a, d0v, totalv);
        else hipLaunchKernelGGL((aux_copy_runs_kernel<float, 2>), dim3(grid_for(totalv)), dim3(256), 0, s, a, d0v, totalv);
        return (int)hipGetLastError();
    }
    if (c->precision == MIFFT_F64) hipLaunchKernelGGL(aux_copy_kernel<double>, dim3(grid_for(a.total)), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(aux_copy_kernel<float>, dim3(grid_for(a.total)), dim3(256), 0, s, a);
    return (int)hipGetLastError();
}

extern "C" int mifft_aux_mul_rows_launch(int f64, void* a, const void* b, long long rows, long long n, hipStream_t s) {
    const long long total = rows * n;
    if (total <= 0) return 0;
    if (f64) hipLaunchKernelGGL(aux_mul_rows_kernel<double>, dim3(grid_for(total)), dim3(256), 0, s, (mifft::cplx<double>*)a, (const mifft::cplx<double>*)b, total, n);
    else hipLaunchKernelGGL(aux_mul_rows_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, (mifft::cplx<float>*)a, (const mifft::cplx<float>*)b, total, n);
    return (int)hipGetLastError();
}

// Verification helper (round 6): how many 16-byte words of two device buffers differ.  A periodic data set must transform into a periodic
// result wherever an item lies in a 256 GiB buffer; comparing every item with its period-mate on the device reads the buffer at the
// streaming rate instead of copying it to the host.  Each wave adds its count to *count (device-accessible, e.g. pinned host memory).
__global__ void __launch_bounds__(256) aux_mismatch_kernel(const uint4* __restrict__ a, const uint4* __restrict__ b, unsigned long long words,
                                                          unsigned long long* count) {
    unsigned long long bad = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256u + threadIdx.x; i < words; i += (unsigned long long)gridDim.x * 256u) {
        const uint4 x = a[i], y = b[i];
        bad += (x.x != y.x) | (x.y != y.y) | (x.z != y.z) | (x.w != y.w);
    }
    for (int off = 32; off > 0; off >>= 1) bad += __shfl_down(bad, off, 64);
    if ((threadIdx.x & 63) == 0 && bad) atomicAdd(count, bad);
}

extern "C" int mifft_aux_mismatch_launch(const void* a, const void* b, unsigned long long words, unsigned long long* count, hipStream_t s) {
    if (words == 0) return 0;
    const unsigned long long want = (words + 255u) / 256u;
    const unsigned grid = (unsigned)(want < 16384ull ? want : 16384ull);
    hipLaunchKernelGGL(aux_mismatch_kernel, dim3(grid), dim3(256), 0, s, (const uint4*)a, (const uint4*)b, words, count);
    return (int)hipGetLastError();
}

// Zero `words16` 16-byte words at p (256-byte aligned counter sets of the persistent launches).  A captured persistent launch zeroes its
// counter set with THIS kernel, not with hipMemsetAsync: graphs whose memset node preceded the kernel stopped zeroing -- every replay
// after it found exhausted tickets and wrote nothing -- once the process built another plan, under the HIP runtime that PyTorch 2.10
// bundles (kernel nodes of the same graphs kept working; tools/r06_runs/m_graph_vs_new_plan.py).
__global__ void __launch_bounds__(256) aux_zero_kernel(uint4* __restrict__ p, unsigned long long words16) {
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256u + threadIdx.x; i < words16; i += (unsigned long long)gridDim.x * 256u)
        p[i] = make_uint4(0u, 0u, 0u, 0u);
}

extern "C" int mifft_aux_zero_launch(void* p, unsigned long long nbytes, hipStream_t s) {
    const unsigned long long words16 = nbytes / 16;
    if (words16 == 0) return 0;
    const unsigned long long want = (words16 + 255u) / 256u;
    hipLaunchKernelGGL(aux_zero_kernel, dim3((unsigned)(want < 1024ull ? want : 1024ull)), dim3(256), 0, s, (uint4*)p, words16);
    return (int)hipGetLastError();
}


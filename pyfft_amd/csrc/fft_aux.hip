// Auxiliary element-wise kernels behind the extensions the reference lists as TODO (TODO.txt:6-8): a general strided complex
// copy (gathers the tiles of a bigger array / the lines of one axis into dense rows and scatters them back; optional
// zero padding, conjugation, a per-position complex multiplier and a scale: the chirp steps of Bluestein's algorithm) and the
// row-wise spectrum product.  One thread per element, fastest index along the threads; these passes are plain streaming
// copies (16 bytes per element and side) and are not on the power-of-two hot path.
#include <hip/hip_runtime.h>
#include "../../include/mifft.h"
#include "fft_butterfly.hpp"

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

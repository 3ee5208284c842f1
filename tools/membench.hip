// Memory-system microbenchmark for MI355X: streaming read / write / copy bandwidth as a function of the
// footprint, to learn (a) the HBM ceilings and (b) whether the 256 MiB Infinity Cache serves re-reads
// and absorbs re-writes of a small scratch buffer.  Development tool.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void k_read(const f4* __restrict__ a, size_t n, f4* sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    f4 acc = {0, 0, 0, 0};
    for (; i < n; i += stride) acc += a[i];
    if (acc.x == 123.456f) sink[0] = acc;
}
__global__ void k_write(f4* __restrict__ a, size_t n, float v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    f4 val = {v, v, v, v};
    for (; i < n; i += stride) a[i] = val;
}
__global__ void k_copy(const f4* __restrict__ a, f4* __restrict__ b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) b[i] = a[i];
}
// tile-structured copy: each block copies a contiguous 32 KiB tile (like an FFT tile kernel without the math)
__global__ void k_copy_tile(const f4* __restrict__ a, f4* __restrict__ b, size_t ntiles) {
    for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const f4* s = a + t * 2048; f4* d = b + t * 2048;
        f4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = s[k * 256 + threadIdx.x];
#pragma unroll
        for (int k = 0; k < 8; ++k) d[k * 256 + threadIdx.x] = v[k];
    }
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t MAXB = (size_t)8 << 30;
    char *A, *B; CK(hipMalloc(&A, MAXB)); CK(hipMalloc(&B, MAXB));
    CK(hipMemset(A, 1, MAXB)); CK(hipMemset(B, 2, MAXB));
    f4* sink; CK(hipMalloc(&sink, 64));
    const int grid = 256 * 8, blk = 256;
    auto timeit = [&](auto fn, int reps) { fn(); CK(hipStreamSynchronize(st)); CK(hipEventRecord(e0, st)); for (int r = 0; r < reps; ++r) fn(); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps; };
    size_t sizes[] = {16, 32, 64, 128, 192, 256, 384, 512, 1024, 4096};
    printf("%8s %12s %12s %12s %12s\n", "MiB", "read GB/s", "write GB/s", "copy GB/s(r+w)", "tilecopy");
    for (size_t mib : sizes) {
        size_t bytes = mib << 20, n = bytes / 16;
        int reps = (int)(((size_t)16 << 30) / bytes); if (reps < 4) reps = 4; if (reps > 400) reps = 400;
        float tr = timeit([&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(blk), 0, st, (const f4*)A, n, sink); }, reps);
        float tw = timeit([&] { hipLaunchKernelGGL(k_write, dim3(grid), dim3(blk), 0, st, (f4*)A, n, 1.0f); }, reps);
        float tc = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(blk), 0, st, (const f4*)A, (f4*)B, n); }, reps);
        float tt = timeit([&] { hipLaunchKernelGGL(k_copy_tile, dim3(grid), dim3(blk), 0, st, (const f4*)A, (f4*)B, bytes / 32768); }, reps);
        printf("%8zu %12.0f %12.0f %12.0f %12.0f\n", mib, bytes / tr / 1e6, bytes / tw / 1e6, 2.0 * bytes / tc / 1e6, 2.0 * bytes / tt / 1e6);
    }
    // scratch experiment: A(big, streamed) -> X(small scratch) -> B(big): two kernels per chunk, X reused.
    printf("\nchunked 2-step copy A->X->B over 4 GiB, X = scratch of given size (reused every chunk)\n");
    printf("%8s %14s %16s\n", "X MiB", "ms total", "GB/s alg(2*4GiB)");
    char* X; CK(hipMalloc(&X, (size_t)4 << 30));
    size_t total = (size_t)4 << 30;
    for (size_t mib : {8, 16, 32, 64, 96, 128, 192, 256, 512, 4096}) {
        size_t cb = mib << 20, n = cb / 16; size_t nch = total / cb;
        auto fn = [&] { for (size_t c = 0; c < nch; ++c) {
            hipLaunchKernelGGL(k_copy, dim3(grid), dim3(blk), 0, st, (const f4*)(A + c * cb), (f4*)X, n);
            hipLaunchKernelGGL(k_copy, dim3(grid), dim3(blk), 0, st, (const f4*)X, (f4*)(B + c * cb), n); } };
        float t = timeit(fn, 3);
        printf("%8zu %14.3f %16.0f\n", mib, t, 2.0 * total / t / 1e6);
    }
    // same but scratch not reused (distinct region per chunk) for comparison at chunk 64 MiB
    {
        size_t cb = (size_t)64 << 20, n = cb / 16, nch = total / cb;
        auto fn = [&] { for (size_t c = 0; c < nch; ++c) {
            hipLaunchKernelGGL(k_copy, dim3(grid), dim3(blk), 0, st, (const f4*)(A + c * cb), (f4*)(X + c * cb), n);
            hipLaunchKernelGGL(k_copy, dim3(grid), dim3(blk), 0, st, (const f4*)(X + c * cb), (f4*)(B + c * cb), n); } };
        float t = timeit(fn, 3);
        printf("no-reuse 64MiB chunks: %.3f ms %.0f GB/s alg\n", t, 2.0 * total / t / 1e6);
    }
    // launch overhead: empty-ish kernels back to back
    {
        float t = timeit([&] { for (int i = 0; i < 1000; ++i) hipLaunchKernelGGL(k_write, dim3(256), dim3(256), 0, st, (f4*)A, (size_t)65536, 1.0f); }, 3);
        printf("1000 small dependent launches: %.3f ms (%.2f us each)\n", t, t);
    }
    return 0;
}

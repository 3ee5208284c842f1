// Column-tile access-pattern microbenchmark: copies [outer][R][C] c64 matrices tile by tile where a tile is
// R rows x W columns (W*8 bytes per row segment, row stride C*8 bytes), with the same thread count / tile
// geometry options as the FFT COL kernels but no math.  Tells how much of the COL kernels' deficit is the
// memory pattern and occupancy rather than the FFT itself.  Development tool.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

// each thread moves PPT/2 16-byte pieces; tile = R x W complex; optional LDS round trip with barriers
template <int R, int W, int NT, bool TRANSPOSE_OUT, bool USE_LDS, int SWZ>
__global__ void __launch_bounds__(NT) k_coltile(const f4* __restrict__ a, f4* __restrict__ b, int C, long long ntiles_per_mat) {
    constexpr int P = R * W, PPT = P / NT, NV = PPT / 2, WV = W / 2;  // WV = 16B vectors per row segment
    __shared__ f4 lds[USE_LDS ? P / 2 : 1];
    long long t = blockIdx.x;
    if (SWZ) {  // put SWZ consecutive tiles on the same XCD (blocks b and b+8 share an XCD)
        long long x = t % 8, i = t / 8;
        t = (i / SWZ) * (8 * SWZ) + x * SWZ + (i % SWZ);
    }
    const long long mat = t / ntiles_per_mat, ct = t % ntiles_per_mat;
    const f4* src = a + mat * ((long long)R * C / 2) + ct * WV;
    f4* dst = b + mat * ((long long)R * C / 2);
    f4 v[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        int e = k * NT + threadIdx.x;
        int r = e / WV, c = e % WV;
        v[k] = src[(long long)r * (C / 2) + c];
    }
    if (USE_LDS) {
#pragma unroll
        for (int k = 0; k < NV; ++k) lds[k * NT + threadIdx.x] = v[k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NV; ++k) v[k] = lds[(k * NT + threadIdx.x) ^ 1];
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        int e = k * NT + threadIdx.x;
        if (TRANSPOSE_OUT) {  // W rows of R contiguous points: row (ct*W + c2), like the Stockham pass-0 write
            int c2 = e / (R / 2), q = e % (R / 2);
            dst[(ct * W + c2) * (long long)(R / 2) + q] = v[k];
        } else {
            int r = e / WV, c = e % WV;
            dst[(long long)r * (C / 2) + ct * WV + c] = v[k];
        }
    }
}

template <int R, int W, int NT, bool TR, bool USE_LDS, int SWZ>
void run(const char* name, const f4* A, f4* B, int C, long long nmat, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
    long long tpm = C / W, ntiles = nmat * tpm;
    auto fn = [&] { hipLaunchKernelGGL((k_coltile<R, W, NT, TR, USE_LDS, SWZ>), dim3((unsigned)ntiles), dim3(NT), 0, st, A, B, C, tpm); };
    fn(); CK(hipStreamSynchronize(st));
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, st)); for (int i = 0; i < 4; ++i) fn(); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms / 4 < best) best = ms / 4;
    }
    double bytes = 2.0 * nmat * R * (double)C * 8;
    printf("%-44s R=%4d W=%3d NT=%4d  %.3f ms  %7.0f GB/s (r+w)\n", name, R, W, NT, best, bytes / best / 1e6);
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const long long nmat = 256;  // 256 matrices of 1024x1024 c64 = 2 GiB
    size_t bytes = (size_t)nmat * 1024 * 1024 * 8;
    f4 *A, *B; CK(hipMalloc(&A, bytes)); CK(hipMalloc(&B, bytes)); CK(hipMemset(A, 1, bytes)); CK(hipMemset(B, 0, bytes));
    printf("matrix = [1024 rows][1024 cols] c64, %lld matrices\n", nmat);
    run<1024, 16, 1024, false, false, 0>("col 1024x16 regs only, in-place pattern", A, B, 1024, nmat, st, e0, e1);
    run<1024, 16, 1024, false, true, 0>("col 1024x16 +LDS(128K), in-place pattern", A, B, 1024, nmat, st, e0, e1);
    run<1024, 16, 1024, true, true, 0>("col 1024x16 +LDS, transposed write", A, B, 1024, nmat, st, e0, e1);
    run<1024, 16, 512, false, false, 0>("col 1024x16 NT512 regs only", A, B, 1024, nmat, st, e0, e1);
    run<1024, 8, 512, false, false, 0>("col 1024x8 regs only", A, B, 1024, nmat, st, e0, e1);
    run<1024, 8, 512, false, true, 0>("col 1024x8 +LDS(64K)", A, B, 1024, nmat, st, e0, e1);
    run<1024, 8, 512, false, true, 2>("col 1024x8 +LDS(64K) xcd-pair swizzle", A, B, 1024, nmat, st, e0, e1);
    run<1024, 8, 512, true, true, 0>("col 1024x8 +LDS transposed write", A, B, 1024, nmat, st, e0, e1);
    run<1024, 4, 256, false, true, 0>("col 1024x4 +LDS(32K)", A, B, 1024, nmat, st, e0, e1);
    run<1024, 4, 256, false, true, 4>("col 1024x4 +LDS(32K) xcd-quad swizzle", A, B, 1024, nmat, st, e0, e1);
    run<1024, 32, 1024, false, false, 0>("col 1024x32 regs only (NT1024, 32pt/thr)", A, B, 1024, nmat, st, e0, e1);
    run<256, 16, 256, false, true, 0>("col 256x16 +LDS(32K) [rows 0..255 only]", A, B, 1024, nmat, st, e0, e1);
    run<256, 32, 512, false, true, 0>("col 256x32 +LDS(64K)", A, B, 1024, nmat, st, e0, e1);
    run<128, 32, 256, false, true, 0>("col 128x32 +LDS(32K)", A, B, 1024, nmat, st, e0, e1);
    run<64, 64, 256, false, true, 0>("col 64x64 +LDS(32K)", A, B, 1024, nmat, st, e0, e1);
    run<4, 1024, 256, false, true, 0>("row-like 4x1024 +LDS(32K)", A, B, 1024, nmat, st, e0, e1);
    return 0;
}

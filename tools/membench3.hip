// Access-shape check for the planned 64-points-per-thread COL kernel: 256 threads per 1024x16 tile, every
// thread moves 64 points with 8-byte accesses (rows a*256 + b1*16 + b0, lane = (b0, c) with c fastest),
// optional LDS bounce of 64 KiB so that two work-groups share a CU.  Pure copy, no math.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE, int LDSK>  // MODE 0: in-place pattern store, 1: transposed store (16B pairs, u fastest)
__global__ void __launch_bounds__(256) k_p(const f2* __restrict__ a, f2* __restrict__ b, long long tiles_per_mat) {
    __shared__ f2 lds[LDSK * 128];  // LDSK KiB
    const long long t = blockIdx.x, mat = t / tiles_per_mat, ct = t % tiles_per_mat;
    const int tid = threadIdx.x, c = tid & 15, b0 = tid >> 4;
    const f2* src = a + mat * (1024ll * 1024) + ct * 16 + c;
    f2 v[64];
#pragma unroll
    for (int k = 0; k < 64; ++k) {  // k = a*16 + b1
        int row = (k >> 4) * 256 + (k & 15) * 16 + b0;
        v[k] = src[(long long)row * 1024];
    }
    // bounce through LDS in 4 rounds of 16 values (keeps the register pressure / barrier structure realistic)
#pragma unroll
    for (int rd = 0; rd < 4; ++rd) {
        f2* buf = lds + (rd & 1) * (LDSK * 64);
#pragma unroll
        for (int k = 0; k < 16; ++k) buf[(k * 256 + tid) % (LDSK * 64)] = v[rd * 16 + k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) v[rd * 16 + k] = buf[((k * 256 + tid) ^ 16) % (LDSK * 64)];
    }
    if (MODE == 0) {
        f2* dst = b + mat * (1024ll * 1024) + ct * 16 + c;
#pragma unroll
        for (int k = 0; k < 64; ++k) {  // k = qb0*4 + qa ; u = b0 ; q = qb0*64 + u*4 + qa
            int q = (k >> 2) * 64 + b0 * 4 + (k & 3);
            dst[(long long)q * 1024] = v[k];
        }
    } else {
        // lanes: u fastest (tid & 15), column = tid >> 4 ; out[(ct*16 + col)][q], q = qb0*64 + u*4 + qa
        const int u = tid & 15, col = tid >> 4;
        f4* dst = (f4*)(b + mat * (1024ll * 1024) + (ct * 16 + col) * 1024ll);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            f4 x = {v[4 * k].x, v[4 * k].y, v[4 * k + 1].x, v[4 * k + 1].y};
            f4 y = {v[4 * k + 2].x, v[4 * k + 2].y, v[4 * k + 3].x, v[4 * k + 3].y};
            dst[(k * 64 + u * 4) / 2] = x;
            dst[(k * 64 + u * 4) / 2 + 1] = y;
        }
    }
}

template <int MODE, int LDSK> void run(const char* name, const f2* A, f2* B, long long nmat, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
    long long tpm = 64, ntiles = nmat * tpm;
    auto fn = [&] { hipLaunchKernelGGL((k_p<MODE, LDSK>), dim3((unsigned)ntiles), dim3(256), 0, st, A, B, tpm); };
    fn(); CK(hipStreamSynchronize(st));
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, st)); for (int i = 0; i < 4; ++i) fn(); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms / 4 < best) best = ms / 4;
    }
    double bytes = 2.0 * nmat * 1024.0 * 1024 * 8;
    printf("%-50s %.3f ms  %7.0f GB/s (r+w)\n", name, best, bytes / best / 1e6);
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const long long nmat = 512;
    size_t bytes = (size_t)nmat * 1024 * 1024 * 8;
    f2 *A, *B; CK(hipMalloc(&A, bytes)); CK(hipMalloc(&B, bytes)); CK(hipMemset(A, 1, bytes)); CK(hipMemset(B, 0, bytes));
    run<0, 64>("P-shape 256thr x 64pt, LDS 64K (2 WG/CU), in-place", A, B, nmat, st, e0, e1);
    run<1, 64>("P-shape 256thr x 64pt, LDS 64K (2 WG/CU), transposed", A, B, nmat, st, e0, e1);
    run<0, 32>("P-shape 256thr x 64pt, LDS 32K (4 WG/CU), in-place", A, B, nmat, st, e0, e1);
    run<1, 32>("P-shape 256thr x 64pt, LDS 32K (4 WG/CU), transposed", A, B, nmat, st, e0, e1);
    run<0, 128>("P-shape 256thr x 64pt, LDS 128K (1 WG/CU), in-place", A, B, nmat, st, e0, e1);
    return 0;
}

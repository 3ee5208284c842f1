// Access-shape check #2: NT threads per 1024x16 tile, PPT = 16384/NT points per thread with 8-byte accesses,
// LDS bounce sized so that `WGS` work-groups fit a CU.  Pure copy, no math.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

template <int NT, int LDSK, int MINW>
__global__ void __launch_bounds__(NT, MINW) k_p(const f2* __restrict__ a, f2* __restrict__ b, long long tiles_per_mat) {
    constexpr int PPT = 16384 / NT, NB0 = NT / 16;  // rows handled per load instruction across the WG
    __shared__ f2 lds[LDSK * 128];
    const long long t = blockIdx.x, mat = t / tiles_per_mat, ct = t % tiles_per_mat;
    const int tid = threadIdx.x, c = tid & 15, b0 = tid >> 4;
    const f2* src = a + mat * (1024ll * 1024) + ct * 16 + c;
    f2 v[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) v[k] = src[(long long)(k * NB0 + b0) * 1024];
    // bounce every value through LDS without aliasing: rounds of RV values per thread, single buffer
    constexpr int E = LDSK * 128, RV = (E / NT) < PPT ? (E / NT) : PPT, RND = PPT / RV;
#pragma unroll
    for (int rd = 0; rd < RND; ++rd) {
#pragma unroll
        for (int k = 0; k < RV; ++k) lds[k * NT + tid] = v[rd * RV + k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < RV; ++k) v[rd * RV + k] = lds[(k * NT + tid) ^ 16];
        __syncthreads();
    }
    f2* dst = b + mat * (1024ll * 1024) + ct * 16 + c;
#pragma unroll
    for (int k = 0; k < PPT; ++k) dst[(long long)(k * NB0 + b0) * 1024] = v[k];
}

template <int NT, int LDSK, int MINW> void run(const char* name, const f2* A, f2* B, long long nmat, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
    long long tpm = 64, ntiles = nmat * tpm;
    auto fn = [&] { hipLaunchKernelGGL((k_p<NT, LDSK, MINW>), dim3((unsigned)ntiles), dim3(NT), 0, st, A, B, tpm); };
    fn(); CK(hipStreamSynchronize(st));
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, st)); for (int i = 0; i < 4; ++i) fn(); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms / 4 < best) best = ms / 4;
    }
    double bytes = 2.0 * nmat * 1024.0 * 1024 * 8;
    printf("%-56s %.3f ms  %7.0f GB/s (r+w)\n", name, best, bytes / best / 1e6);
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const long long nmat = 512;
    size_t bytes = (size_t)nmat * 1024 * 1024 * 8;
    f2 *A, *B; CK(hipMalloc(&A, bytes)); CK(hipMalloc(&B, bytes)); CK(hipMemset(A, 1, bytes)); CK(hipMemset(B, 0, bytes));
    run<512, 64, 4>("512thr x 32pt, LDS 64K, <=128 VGPR (2 WG/CU, 16 waves)", A, B, nmat, st, e0, e1);
    run<512, 32, 4>("512thr x 32pt, LDS 32K, <=128 VGPR (2 WG/CU by VGPR)", A, B, nmat, st, e0, e1);
    run<512, 32, 6>("512thr x 32pt, LDS 32K, <=80 VGPR  (3 WG/CU, 24 waves)", A, B, nmat, st, e0, e1);
    run<1024, 128, 4>("1024thr x 16pt, LDS 128K (1 WG/CU, 16 waves)", A, B, nmat, st, e0, e1);
    run<1024, 64, 8>("1024thr x 16pt, LDS 64K, <=64 VGPR (2 WG/CU, 32 waves)", A, B, nmat, st, e0, e1);
    run<256, 32, 4>("256thr x 64pt, LDS 32K, <=128 VGPR (4 WG/CU, 16 waves)", A, B, nmat, st, e0, e1);
    run<256, 32, 3>("256thr x 64pt, LDS 32K, <=168 VGPR (3 WG/CU, 12 waves)", A, B, nmat, st, e0, e1);
    run<256, 64, 2>("256thr x 64pt, LDS 64K, <=256 VGPR (2 WG/CU, 8 waves)", A, B, nmat, st, e0, e1);
    return 0;
}

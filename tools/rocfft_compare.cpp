// Vendor-library comparator (counterpart of the reference's cuda/test.cu, which times CUFFT on the doc's
// shapes): times hipFFT/rocFFT batched c2c transforms, out of place, on the BASELINE shapes.  Comparator only.
// build: hipcc -O2 tools/rocfft_compare.cpp -o tools/rocfft_compare -lhipfft
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { auto e = (x); if (e != 0) { printf("error %d at line %d\n", (int)e, __LINE__); return 1; } } while (0)

static int run(const char* name, int rank, int* dims, long long batch, bool dp) {
    long long n = 1; double lg = 0;
    for (int i = 0; i < rank; ++i) { n *= dims[i]; lg += std::log2((double)dims[i]); }
    size_t esz = dp ? 16 : 8, bytes = (size_t)n * batch * esz;
    void *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    std::vector<float> h(1 << 22); for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
    for (size_t off = 0; off < bytes; off += h.size() * 4) CK(hipMemcpy((char*)a + off, h.data(), std::min(bytes - off, h.size() * 4), hipMemcpyHostToDevice));
    hipfftHandle plan;
    CK(hipfftPlanMany(&plan, rank, dims, nullptr, 1, (int)n, nullptr, 1, (int)n, dp ? HIPFFT_Z2Z : HIPFFT_C2C, (int)batch));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto exec = [&]() { return dp ? hipfftExecZ2Z(plan, (hipfftDoubleComplex*)a, (hipfftDoubleComplex*)b, HIPFFT_FORWARD)
                                  : hipfftExecC2C(plan, (hipfftComplex*)a, (hipfftComplex*)b, HIPFFT_FORWARD); };
    CK(exec()); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0)); for (int i = 0; i < 5; ++i) CK(exec()); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms / 5 < best) best = ms / 5;
    }
    double alg = 2.0 * n * batch * esz;
    printf("%-28s batch %-7lld %s  %.3f ms  %.1f GB/s alg (%.1f%% of 8 TB/s)  %.0f GFLOPS\n", name, batch, dp ? "z2z" : "c2c", best,
           alg / best / 1e6, alg / best / 1e6 / 80.0, 5.0 * n * lg * batch / best / 1e6);
    hipfftDestroy(plan); hipFree(a); hipFree(b);
    return 0;
}

int main() {
    int d1[] = {1024}; run("1D 1024", 1, d1, 65536, false);
    int d1b[] = {4096}; run("1D 4096", 1, d1b, 16384, false);
    int d1c[] = {65536}; run("1D 2^16", 1, d1c, 4096, false);
    int d2[] = {1 << 20}; run("1D 2^20 (config 2 shape)", 1, d2, 512, false);
    int d3[] = {1024, 1024}; run("2D 1024x1024 (config 3)", 2, d3, 256, false);
    int d4[] = {256, 256, 256}; run("3D 256^3 (config 4)", 3, d4, 8, true);
    int d5[] = {1 << 22}; run("1D 2^22 (config 5 shape)", 1, d5, 64, false);
    int d6[] = {128, 128, 128}; run("3D 128^3", 3, d6, 64, false);
    return 0;
}

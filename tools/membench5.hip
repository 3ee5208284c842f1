// Access-pattern ceiling of the pass-pair YZ tiles (round 3): an in-place copy (load every point of a tile, store it back) with the
// tile geometries of csrc/fft_pair.hpp on a 256^3 fp64 batch -- how fast can ANY kernel stream these gathers?
//   A  W = 8 columns (128-byte segments), r stride 32 rows, 8 r x 256 z   (the YZ tile of the 32 x 8 split: 2048 segments)
//   B  W = 16 columns (256-byte segments), r stride 64 rows, 4 r x 256 z  (a 16-column tile of the 64 x 4 split: 1024 segments)
//   C  contiguous 256 KiB blocks                                          (reference)
// Build: hipcc -O3 --offload-arch=gfx950 tools/membench5.hip -o tools/membench5
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

template <int MODE> __global__ void __launch_bounds__(1024) k_tile(d2* p, long long tiles) {
    const int t = threadIdx.x;
    const long long tile = blockIdx.x;
    if (tile >= tiles) return;
    d2 v[16];
    long long off[16];
    if (MODE == 0) {          // A
        const long long xf = tile / 1024, rem = tile % 1024, c = rem % 32, q = rem / 32;
        const int x = t & 7, zz = t >> 3;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int k = 0; k < 8; ++k) off[b * 8 + k] = xf * 16777216ll + (long long)(b * 128 + zz) * 65536 + (k * 32 + q) * 256 + c * 8 + x;
    } else if (MODE == 1) {   // B
        const long long xf = tile / 1024, rem = tile % 1024, c = rem % 16, q = rem / 16;
        const int x = t & 15, zz = t >> 4;
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int k = 0; k < 4; ++k) off[b * 4 + k] = xf * 16777216ll + (long long)(b * 64 + zz) * 65536 + (k * 64 + q) * 256 + c * 16 + x;
    } else {                  // C
#pragma unroll
        for (int i = 0; i < 16; ++i) off[i] = tile * 16384 + i * 1024 + t;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = p[off[i]];
#pragma unroll
    for (int i = 0; i < 16; ++i) { v[i].x += 1.0; p[off[i]] = v[i]; }
}

int main() {
    const long long batch = 16, n = 16777216ll * batch;
    d2* p;
    CK(hipMalloc(&p, n * 16));
    CK(hipMemset(p, 0, n * 16));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const long long tiles = n / 16384;
    const char* names[3] = {"A: 8 columns x (8 r, stride 32 rows) x 256 z", "B: 16 columns x (4 r, stride 64 rows) x 256 z", "C: contiguous 256 KiB"};
    for (int mode = 0; mode < 3; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0, 0));
            if (mode == 0) hipLaunchKernelGGL(k_tile<0>, dim3((unsigned)tiles), dim3(1024), 0, 0, p, tiles);
            else if (mode == 1) hipLaunchKernelGGL(k_tile<1>, dim3((unsigned)tiles), dim3(1024), 0, 0, p, tiles);
            else hipLaunchKernelGGL(k_tile<2>, dim3((unsigned)tiles), dim3(1024), 0, 0, p, tiles);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("%-52s %.3f ms  %.2f TB/s (read + write)\n", names[mode], best, 2.0 * n * 16 / best / 1e9);
    }
    return 0;
}

// Instances of the wave-autonomous small-transform kernels (fft_wave.hpp): 1-D N = 4 ... 32 (fp32), 2 ... 16 (fp64) and the
// (16, 16) fp32 plane; interleaved data, dense rows.
#include "mifft_internal.h"
#include "fft_wave.hpp"

namespace {
template <typename T, int N> int launch(const mifft::WaveArgs* a, int max_blocks, hipStream_t s) {
    constexpr int G = N * (int)sizeof(mifft::cplx<T>) / 16;
    constexpr int U = G >= 8 ? 1 : 8 / G;   // chunks per wave and step (fft_wave.hpp)
    const long long chunks = (a->pieces + 64 * G - 1) / (64 * G);
    long long blocks = (chunks + 4 * U - 1) / (4 * U);
    if (blocks > max_blocks) blocks = max_blocks;
    if (blocks < 1) return 0;
    hipLaunchKernelGGL((mifft::fft_wave_kernel<T, N>), dim3((unsigned)blocks), dim3(256), 0, s, *a);
    return (int)hipGetLastError();
}
}  // namespace

extern "C" int mifft_wave_supported(int f64, int N) {
    if (f64) return (N == 2 || N == 4 || N == 8 || N == 16) ? 0 : -2;
    return (N == 4 || N == 8 || N == 16 || N == 32) ? 0 : -2;
}

// max_blocks: cap of the grid-stride launch (the caller passes a few work-groups per CU)
extern "C" int mifft_wave_launch(int f64, int N, const mifft::WaveArgs* a, int max_blocks, hipStream_t s) {
    if (f64) {
        switch (N) {
            case 2: return launch<double, 2>(a, max_blocks, s);
            case 4: return launch<double, 4>(a, max_blocks, s);
            case 8: return launch<double, 8>(a, max_blocks, s);
            case 16: return launch<double, 16>(a, max_blocks, s);
        }
    } else {
        switch (N) {
            case 4: return launch<float, 4>(a, max_blocks, s);
            case 8: return launch<float, 8>(a, max_blocks, s);
            case 16: return launch<float, 16>(a, max_blocks, s);
            case 32: return launch<float, 32>(a, max_blocks, s);
        }
    }
    return -2;
}

extern "C" int mifft_wave_16x16_launch(const mifft::WaveArgs* a, int max_blocks, hipStream_t s) {
    const long long chunks = (a->pieces / 128 + 7) / 8;
    long long blocks = (chunks + 3) / 4;
    if (blocks > max_blocks) blocks = max_blocks;
    if (blocks < 1) return 0;
    hipLaunchKernelGGL((mifft::fft_wave_16x16_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, s, *a);
    return (int)hipGetLastError();
}

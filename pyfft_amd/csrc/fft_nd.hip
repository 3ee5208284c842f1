// Instances of the in-LDS N-D kernel (fft_nd.hpp).  Tile = the smallest of 4096 / 8192 / 16384 points (fp32; half
// of that in fp64) that holds one transform: small tiles keep 4 work-groups per CU, the largest one (128 KiB of LDS,
// 1024 threads) still beats one HBM round trip per axis.
#include "mifft_internal.h"
#include "fft_nd.hpp"

extern "C" int mifft_nd_max_points(int f64) { return f64 ? 8192 : 16384; }

namespace {
template <typename T, int P, int NT> int launch(const mifft::NdArgs* a, hipStream_t s) {
    const long long tiles = (a->total + P - 1) / P;
    if (tiles <= 0) return 0;
    if (tiles > 2147483647ll) return -1;
    hipLaunchKernelGGL((mifft::fft_nd_kernel<T, P, NT>), dim3((unsigned)tiles), dim3(NT), 0, s, *a);
    return (int)hipGetLastError();
}
}  // namespace

// n = points of one transform (x*y*z)
extern "C" int mifft_nd_launch(int f64, long long n, const mifft::NdArgs* a, hipStream_t s) {
    if (f64) {
        if (n <= 2048) return launch<double, 2048, 256>(a, s);
        if (n <= 4096) return launch<double, 4096, 512>(a, s);
        return launch<double, 8192, 1024>(a, s);
    }
    if (n <= 4096) return launch<float, 4096, 256>(a, s);
    if (n <= 8192) return launch<float, 8192, 512>(a, s);
    return launch<float, 16384, 1024>(a, s);
}

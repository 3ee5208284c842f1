// Instances of the in-LDS N-D kernel (fft_nd.hpp): fp32 tiles of 4096 points, fp64 tiles of 2048 points.
#include "mifft_internal.h"
#include "fft_nd.hpp"

extern "C" int mifft_nd_max_points(int f64) { return f64 ? 2048 : 4096; }

extern "C" int mifft_nd_launch(int f64, const mifft::NdArgs* a, hipStream_t s) {
    const long long P = f64 ? 2048 : 4096;
    const long long tiles = (a->total + P - 1) / P;
    if (tiles <= 0) return 0;
    if (tiles > 2147483647ll) return -1;
    if (f64)
        hipLaunchKernelGGL((mifft::fft_nd_kernel<double, 2048, 256>), dim3((unsigned)tiles), dim3(256), 0, s, *a);
    else
        hipLaunchKernelGGL((mifft::fft_nd_kernel<float, 4096, 256>), dim3((unsigned)tiles), dim3(256), 0, s, *a);
    return (int)hipGetLastError();
}

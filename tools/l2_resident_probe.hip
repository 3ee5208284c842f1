// Does an exchange buffer stay resident in an XCD's 4 MiB L2 NEXT TO the HBM streams of the same work-groups?
// (development probe behind the on-XCD hand-off of round 3; results: profiles/r03_a_l2_resident_probe.log)
//
// 512 persistent work-groups (64 per XCD, rank found with HW_REG_XCC_ID).  Per iteration a work-group
//     loads a 32 KiB tile of A                     (stream in,  cache policy LD)
//     stores it into slot (r + it) % slots of its XCD's exchange buffer          (plain stores: stay in L2)
//     loads another slot of that buffer            (sc1 = L1-bypassing, L2-served)
//     stores the sum to a 32 KiB tile of B         (stream out, cache policy ST)
// so the exchange volume equals the stream volume, as in a two-pass transform whose intermediate stays on the XCD, and the
// exchange FOOTPRINT is slots * 32 KiB per XCD.  No flags: the data race is irrelevant to the traffic.  Run under
//     rocprofv3 --pmc FETCH_SIZE   /   --pmc WRITE_SIZE   (separate runs) with --kernel-trace
// and compare per dispatch with the printed algorithmic bytes: FETCH_SIZE * 2 == in  and  WRITE_SIZE == out  mean the
// exchange never left the L2.
// Build: hipcc -O3 --offload-arch=gfx950 tools/l2_resident_probe.hip -o tools/l2_resident_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef unsigned u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7u;
}

// cache policies: 0 plain, 1 nt, 2 sc1, 3 sc0 sc1, 4 sc1 nt, 5 sc0 sc1 nt
template <int P> __device__ __forceinline__ u4 ld(const u4* p) {
    u4 v;
    if constexpr (P == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    else if constexpr (P == 1) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
    else if constexpr (P == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    else if constexpr (P == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    else if constexpr (P == 4) asm volatile("global_load_dwordx4 %0, %1, off sc1 nt" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int P> __device__ __forceinline__ void st(u4* p, u4 v) {
    if constexpr (P == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    else if constexpr (P == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
    else if constexpr (P == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else if constexpr (P == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    else if constexpr (P == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
}

// The loads above are asynchronous and invisible to the compiler: the wait names every destination register as read-write,
// so no use (and no re-allocation of those registers) can be scheduled in front of it.
__device__ __forceinline__ void wait8(u4 (&a)[8]) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])::"memory");
}

struct Args {
    const u4* A;
    u4* B;
    u4* X;          // [8][slots][2048] 16-byte words
    unsigned* cnt;  // [8] census
    unsigned slots, iters, exchange;
};

template <int LD, int ST>
__global__ void __launch_bounds__(256, 2) k_mix(const Args p) {
    __shared__ unsigned s_rank;
    const unsigned x = xcc_id();
    const unsigned tid = threadIdx.x;
    if (tid == 0) s_rank = atomicAdd(&p.cnt[x], 1u);
    __syncthreads();
    const unsigned r = s_rank;
    u4* const xb = p.X + (size_t)x * p.slots * 2048u;
    for (unsigned it = 0; it < p.iters; ++it) {
        const size_t g = ((size_t)(x * 64u + (r & 63u)) * p.iters + it) * 2048u;
        u4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = ld<LD>(p.A + g + k * 256 + tid);
        wait8(v);
        if (p.exchange) {
            u4* const s0 = xb + (size_t)((r + it) % p.slots) * 2048u;
#pragma unroll
            for (int k = 0; k < 8; ++k) st<0>(s0 + k * 256 + tid, v[k]);
            const u4* const s1 = xb + (size_t)((r + it + p.slots / 2u + 1u) % p.slots) * 2048u;
            u4 w[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) w[k] = ld<2>(s1 + k * 256 + tid);
            wait8(w);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] += w[k];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) st<ST>(p.B + g + k * 256 + tid, v[k]);
        asm volatile("s_nop 1" ::: "memory");   // store-data hazard: the next iteration's loads overwrite v[]
    }
}

template <int LD, int ST> void run(Args p, hipStream_t s, const char* name) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemsetAsync(p.cnt, 0, 64, s));
        CK(hipEventRecord(e0, s));
        hipLaunchKernelGGL((k_mix<LD, ST>), dim3(512), dim3(256), 0, s, p);
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double bytes = 512.0 * p.iters * 32768.0;
    printf("k_mix<%d,%d> %-22s exchange=%u footprint %5u KiB/XCD: %.3f ms  stream in+out %.2f TB/s  (in = out = %.1f MiB per dispatch)\n", LD, ST, name,
           p.exchange, p.exchange ? p.slots * 32u : 0u, best, 2.0 * bytes / best / 1e9, bytes / 1048576.0);
    fflush(stdout);
}

int main() {
    hipStream_t s;
    CK(hipStreamCreate(&s));
    Args p;
    p.iters = 64;   // 1 GiB in, 1 GiB out per dispatch
    const size_t bytes = (size_t)512 * p.iters * 32768;
    u4 *A, *B, *X;
    unsigned* cnt;
    CK(hipMalloc(&A, bytes));
    CK(hipMalloc(&B, bytes));
    CK(hipMalloc(&X, (size_t)8 * 128 * 32768));
    CK(hipMalloc(&cnt, 64));
    CK(hipMemset(A, 1, bytes));
    CK(hipMemset(B, 0, bytes));
    CK(hipMemset(X, 0, (size_t)8 * 128 * 32768));
    p.A = A; p.B = B; p.X = X; p.cnt = cnt;
    for (unsigned slots : {0u, 8u, 16u, 32u, 64u, 96u}) {
        p.exchange = slots ? 1u : 0u;
        p.slots = slots ? slots : 1u;
        run<0, 0>(p, s, "ld plain / st plain");
        run<1, 1>(p, s, "ld nt / st nt");
        run<1, 2>(p, s, "ld nt / st sc1");
        run<1, 4>(p, s, "ld nt / st sc1 nt");
        run<4, 4>(p, s, "ld sc1 nt / st sc1 nt");
        run<5, 5>(p, s, "ld sc0sc1nt / st same");
        run<3, 3>(p, s, "ld sc0sc1 / st sc0sc1");
    }
    return 0;
}

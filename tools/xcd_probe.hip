// XCD-level feasibility probe for a single-crossing N = 2^20 transform (development tool; results are kept under
// profiles/r02_xcd_probe*.log).  Three questions, each one number the design of fft_xcd2.hpp depends on:
//   T1  census: how many of the 512 persistent work-groups land on each XCD (HW_REG_XCC_ID)
//   T2  B1: the streaming bandwidth ONE XCD can pull when only k of the 8 XCDs stream (column-tile access
//       pattern of the COL kernels: 256 threads x 64 eight-byte accesses in flight, non-temporal)
//   T3  the price of an all-to-all round among the 64 work-groups of one XCD through that XCD's L2:
//       plain stores -> vmcnt drain -> per-producer flag; consumer polls its 16 producers' flags, then sc1 loads
//       (L1 bypass, L2-served).  2 MiB per XCD and round, double-buffered, read-done flags against overwriting.
// Build: hipcc -O3 --offload-arch=gfx950 tools/xcd_probe.hip -o tools/xcd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <type_traits>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned long long u64;

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15u;
}

__global__ void k_census(unsigned* cnt) {
    if (threadIdx.x == 0) atomicAdd(&cnt[xcc_id()], 1u);
}

// ---- T2: per-XCD streaming copy, only XCDs with id < nactive work; per-XCD tile queue
template <bool READONLY>
__global__ void __launch_bounds__(256, 2) k_stream(const f2* __restrict__ a, f2* __restrict__ b, unsigned* queue,
                                                   unsigned nactive, unsigned tiles_per_xcd, f2* sink) {
    __shared__ unsigned s_tile;
    const unsigned x = xcc_id();
    if (x >= nactive) return;
    const int tid = threadIdx.x, c = tid & 15, b0 = tid >> 4;
    f2 acc = {0.f, 0.f};
    for (;;) {
        __syncthreads();
        if (tid == 0) s_tile = atomicAdd(&queue[x], 1u);
        __syncthreads();
        const unsigned tile = s_tile;
        if (tile >= tiles_per_xcd) break;
        const long long g = (long long)x * tiles_per_xcd + tile, mat = g >> 6, ct = g & 63;
        const f2* src = a + mat * (1024ll * 1024) + ct * 16 + c;
        f2 v[64];
#pragma unroll
        for (int k = 0; k < 64; ++k) v[k] = __builtin_nontemporal_load(src + (long long)(k * 16 + b0) * 1024);
        if (READONLY) {
#pragma unroll
            for (int k = 0; k < 64; ++k) acc += v[k];
        } else {
            f2* dst = b + mat * (1024ll * 1024) + ct * 16 + c;
#pragma unroll
            for (int k = 0; k < 64; ++k) __builtin_nontemporal_store(v[k], dst + (long long)(k * 16 + b0) * 1024);
        }
    }
    if (READONLY && acc.x == 123.456f) sink[0] = acc;
}

// ---- T3: all-to-all rounds inside each XCD
struct XArgs {
    unsigned* cnt;      // [8] census counters (rank assignment)
    unsigned* ready;    // [8][64] rounds written by producer r
    unsigned* rdone;    // [8][64] rounds read by consumer r'
    unsigned* err;      // [0] timeout, [1] data mismatches, [2] bad census
    u64* scratch;       // [8][2][64][16][256] 8-byte values
    unsigned rounds;
};

__device__ __forceinline__ bool poll16(const unsigned* base, unsigned idx, unsigned target, unsigned* err, bool active) {
    // lanes with active==true poll base[idx] >= target (relaxed agent load = sc1: L1 bypass)
    unsigned spins = 0;
    for (;;) {
        bool ok = true;
        if (active) ok = __hip_atomic_load(base + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target;
        if (__all(ok)) return true;
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1u << 20)) { if (threadIdx.x == 0) atomicAdd(err, 1u); return false; }
    }
}

template <bool VERIFY, int NV>
__global__ void __launch_bounds__(256, 2) k_xchg(const XArgs p) {
    __shared__ unsigned s_rank;
    const unsigned x = xcc_id();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_rank = atomicAdd(&p.cnt[x], 1u);
    __syncthreads();
    const unsigned r = s_rank;
    if (r >= 64) { if (tid == 0) atomicAdd(p.err + 2, 1u); return; }
    unsigned* ready = p.ready + x * 64;
    unsigned* rdone = p.rdone + x * 64;
    u64* scr = p.scratch + (size_t)x * (2 * 64 * NV * 256);
    unsigned bad = 0;
    for (unsigned k = 0; k < p.rounds; ++k) {
        u64* buf = scr + (size_t)(k & 1) * (64 * NV * 256);
        const unsigned qa = (r - k) & 3u;   // this round's consumers: r' = i*4 + qa
        // WAR: consumers of this buffer parity must have read round k-2
        if (k >= 2 && wave == 0) poll16(rdone, (lane & 15) * 4 + qa, k - 1, p.err, lane < 16);
        __syncthreads();
        // NV < 16: only producers with (r >> 2) < NV send (slot = r >> 2), so the footprint is 2 * 64 * NV * 2 KiB per XCD
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const unsigned rc = i * 4 + qa;
            const u64 val = ((u64)k << 40) | ((u64)r << 32) | ((u64)rc << 16) | (unsigned)tid;
            if ((r >> 2) < (unsigned)NV) buf[((size_t)rc * NV + (r >> 2)) * 256 + tid] = val;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(ready + r, k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        // RAW: my 16 producers of this round: r = j*4 + ((r' + k) & 3)
        const unsigned pa = (r + k) & 3u;
        if (wave == 0) poll16(ready, (lane & 15) * 4 + pa, k + 1, p.err, lane < 16);
        __syncthreads();
        u64 got[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j)
            got[j] = __hip_atomic_load(buf + ((size_t)r * NV + j) * 256 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (VERIFY) {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const u64 want = ((u64)k << 40) | ((u64)(j * 4 + pa) << 32) | ((u64)r << 16) | (unsigned)tid;
                bad += got[j] != want;
            }
        } else {
            u64 s = 0;
#pragma unroll
            for (int j = 0; j < NV; ++j) s ^= got[j];
            bad += (s == 0x1234567ull);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(rdone + r, k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (bad) atomicAdd(p.err + 1, bad);
}

int main(int argc, char** argv) {
    const bool only_xchg = argc > 1 && atoi(argv[1]) == 3;
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    unsigned* ctl; CK(hipMalloc(&ctl, 1 << 16));
    unsigned h[16];
    // T1
    for (int grid : {256, 512, 1024}) {
        CK(hipMemsetAsync(ctl, 0, 64, st));
        hipLaunchKernelGGL(k_census, dim3(grid), dim3(256), 0, st, ctl);
        CK(hipMemcpyAsync(h, ctl, 64, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
        printf("T1 census grid=%4d:", grid);
        for (int i = 0; i < 8; ++i) printf(" %u", h[i]);
        printf("\n");
    }
    if (!only_xchg) {
        // T2
        const long long nmat = 1024;  // 8 GiB each side
        const size_t bytes = (size_t)nmat * 1024 * 1024 * 8;
        f2 *A, *B; CK(hipMalloc(&A, bytes)); CK(hipMalloc(&B, bytes)); CK(hipMemset(A, 1, bytes)); CK(hipMemset(B, 0, bytes));
        f2* sink = (f2*)(ctl + 1024);
        for (int ro = 0; ro < 2; ++ro)
            for (unsigned nact : {1u, 2u, 4u, 8u}) {
                const unsigned tiles_per_xcd = 128 * 64;  // 128 transforms = 1 GiB per XCD
                float best = 1e9f;
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipMemsetAsync(ctl, 0, 64, st));
                    CK(hipEventRecord(e0, st));
                    if (ro) hipLaunchKernelGGL(k_stream<true>, dim3(512), dim3(256), 0, st, A, B, ctl, nact, tiles_per_xcd, sink);
                    else hipLaunchKernelGGL(k_stream<false>, dim3(512), dim3(256), 0, st, A, B, ctl, nact, tiles_per_xcd, sink);
                    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
                }
                const double moved = (ro ? 1.0 : 2.0) * nact * 128.0 * 8388608.0;
                printf("T2 %s active XCDs=%u: %.3f ms  total %.0f GB/s  per XCD %.0f GB/s\n", ro ? "read-only" : "copy(r+w) ",
                       nact, best, moved / best / 1e6, moved / best / 1e6 / nact);
            }
        CK(hipFree(A)); CK(hipFree(B));
    }
    // T3
    {
        XArgs p;
        p.cnt = ctl; p.ready = ctl + 64; p.rdone = ctl + 64 + 512; p.err = ctl + 64 + 1024;
        const size_t sbytes = (size_t)8 * 2 * 64 * 16 * 256 * 8;  // 32 MiB: 4 MiB per XCD
        CK(hipMalloc(&p.scratch, sbytes)); CK(hipMemset(p.scratch, 0, sbytes));
        auto bench = [&](auto nvc, int verify, unsigned rounds) {
            constexpr int NV = decltype(nvc)::value;
            p.rounds = rounds;
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipMemsetAsync(ctl, 0, (64 + 1024 + 16) * 4, st));
                CK(hipEventRecord(e0, st));
                if (verify) hipLaunchKernelGGL((k_xchg<true, NV>), dim3(512), dim3(256), 0, st, p);
                else hipLaunchKernelGGL((k_xchg<false, NV>), dim3(512), dim3(256), 0, st, p);
                CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            CK(hipMemcpy(h, p.err, 16, hipMemcpyDeviceToHost));
            const double bytes_round = 64.0 * NV * 2048.0 * (NV / 16.0 < 1 ? 1 : 1);   // bytes written per XCD and round (producers with slot < NV)
            printf("T3 xchg NV=%2d (footprint %4.0f KiB/XCD) verify=%d rounds=%5u: %.3f ms  %.2f us/round  stored %.1f MiB in all  timeouts=%u mismatches=%u badcensus=%u\n",
                   NV, 2 * bytes_round / 1024.0, verify, rounds, best, best * 1e3 / rounds, 8.0 * bytes_round * rounds / 1048576.0, h[0], h[1], h[2]);
        };
        bench(std::integral_constant<int, 16>{}, 1, 1024);
        bench(std::integral_constant<int, 16>{}, 0, 4096);
        bench(std::integral_constant<int, 8>{}, 1, 4096);
        bench(std::integral_constant<int, 4>{}, 1, 4096);
        bench(std::integral_constant<int, 2>{}, 1, 4096);
        bench(std::integral_constant<int, 1>{}, 1, 4096);
    }
    return 0;
}

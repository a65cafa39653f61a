// Dev micro-benchmark: why did copy_bw.cpp read 4.43 TB/s for a 16-byte-per-lane copy when MI355X_MICROARCH.md records 6.29 TB/s
// for a float4 copy? Variants of the same 1 GiB -> 1 GiB copy: load / store cache policy (plain vs non-temporal), loop shape
// (grid-stride persistent blocks vs one block per chunk), loads in flight per lane, block size.
//   hipcc -O3 --offload-arch=gfx950 copy_bw2.cpp -o copy_bw2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float f4;

// PERSIST: grid-stride loop over the whole array with `grid` resident blocks; else one block per UNROLL * blockDim elements.
template <int NTL, int NTS, int UNROLL, bool PERSIST>
__global__ void copyk(const f4* __restrict__ a, f4* __restrict__ y, long n) {
    const long bs = blockDim.x;
    if constexpr (PERSIST) {
        const long stride = (long)gridDim.x * bs;
        for (long i = (long)blockIdx.x * bs + threadIdx.x; i < n; i += stride * UNROLL) {
            f4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const long j = i + u * stride;
                if (j < n) v[u] = NTL ? __builtin_nontemporal_load(a + j) : a[j];
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const long j = i + u * stride;
                if (j < n) { if (NTS) __builtin_nontemporal_store(v[u], y + j); else y[j] = v[u]; }
            }
        }
    } else {
        const long base = (long)blockIdx.x * bs * UNROLL + threadIdx.x;
        f4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const long j = base + u * bs;
            if (j < n) v[u] = NTL ? __builtin_nontemporal_load(a + j) : a[j];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const long j = base + u * bs;
            if (j < n) { if (NTS) __builtin_nontemporal_store(v[u], y + j); else y[j] = v[u]; }
        }
    }
}

template <int NTL, int NTS, int UNROLL, bool PERSIST>
static void run(const char* name, const f4* a, f4* y, long n, int threads, int blocks_per_cu) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const long grid = PERSIST ? 256L * blocks_per_cu : (n + (long)threads * UNROLL - 1) / ((long)threads * UNROLL);
    float best = 1e9;
    for (int rep = 0; rep < 6; ++rep) {
        (void)hipEventRecord(e0);
        copyk<NTL, NTS, UNROLL, PERSIST><<<(unsigned)grid, threads>>>(a, y, n);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    printf("%-58s %8.1f us  %.2f TB/s (read + write)\n", name, best * 1e3, n * 32.0 / best / 1e9);
}

int main() {
    const long n = 64L << 20;                 // 64 Mi x 16 B = 1 GiB in, 1 GiB out (4x the 256 MiB Infinity Cache each)
    f4 *a, *y;
    (void)hipMalloc(&a, n * 16); (void)hipMalloc(&y, n * 16);
    (void)hipMemset(a, 1, n * 16); (void)hipMemset(y, 0, n * 16);
    run<1, 0, 4, true>("persistent 16 blk/CU x256, nt load, plain store, 4 in flight (= copy_bw.cpp)", a, y, n, 256, 16);
    run<0, 0, 4, true>("persistent 16 blk/CU x256, plain load, plain store, 4", a, y, n, 256, 16);
    run<0, 1, 4, true>("persistent 16 blk/CU x256, plain load, nt store, 4", a, y, n, 256, 16);
    run<1, 1, 4, true>("persistent 16 blk/CU x256, nt load, nt store, 4", a, y, n, 256, 16);
    run<1, 1, 8, true>("persistent 8 blk/CU x256, nt load, nt store, 8", a, y, n, 256, 8);
    run<0, 0, 1, false>("one block per 256 elements, plain, 1 in flight", a, y, n, 256, 0);
    run<0, 0, 4, false>("one block per 1024 elements, plain, 4 in flight", a, y, n, 256, 0);
    run<0, 1, 4, false>("one block per 1024 elements, plain load, nt store, 4", a, y, n, 256, 0);
    run<1, 1, 4, false>("one block per 1024 elements, nt load, nt store, 4", a, y, n, 256, 0);
    run<0, 0, 8, false>("one block per 2048 elements, plain, 8 in flight", a, y, n, 256, 0);
    run<0, 0, 4, false>("one block per 4096 elements (1024 threads), plain, 4", a, y, n, 1024, 0);
    run<1, 1, 8, false>("one block per 2048 elements, nt load, nt store, 8", a, y, n, 256, 0);
    // hipMemcpyAsync device-to-device for reference
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0); (void)hipMemcpyAsync(y, a, n * 16, hipMemcpyDeviceToDevice, 0); (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (rep > 0 && ms < best) best = ms;
    }
    printf("%-58s %8.1f us  %.2f TB/s (read + write)\n", "hipMemcpyAsync device to device", best * 1e3, n * 32.0 / best / 1e9);
    // smaller footprints: 64 MiB + 64 MiB (Infinity-Cache resident) - what a copy reads when it does NOT reach HBM
    const long m = 4L << 20;
    run<0, 0, 4, false>("64 MiB -> 64 MiB (Infinity Cache resident), plain, 4", a, y, m, 256, 0);
    return 0;
}

// Dev micro-benchmark: what does this MI355X sustain for pure-read, pure-write and 1:1 copy streams with 16 B per lane
// (the access shape of the depthwise / 1x1 epilogue kernels)?  hipcc -O3 --offload-arch=gfx950 copy_bw.cpp -o copy_bw
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int MODE, int UNROLL>   // 0 read, 1 write, 2 copy, 3 read 2 streams + write 1 (conv + residual)
__global__ __launch_bounds__(256) void k(const u32x4* __restrict__ a, const u32x4* __restrict__ b, u32x4* __restrict__ y, long n,
                                         u32x4* sink) {
    const long stride = (long)gridDim.x * 256;
    u32x4 acc = {0, 0, 0, 0};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride * UNROLL) {
        u32x4 v[UNROLL], w[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const long j = i + u * stride;
            if (MODE != 1) v[u] = j < n ? __builtin_nontemporal_load(a + j) : acc;
            if (MODE == 3) w[u] = j < n ? __builtin_nontemporal_load(b + j) : acc;
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const long j = i + u * stride;
            if (MODE == 0) acc ^= v[u];
            if (MODE == 1 && j < n) y[j] = (u32x4){(unsigned)j, 1u, 2u, 3u};
            if (MODE == 2 && j < n) y[j] = v[u];
            if (MODE == 3 && j < n) y[j] = v[u] ^ w[u];
        }
    }
    if (MODE == 0 && acc[0] == 0x12345678u) *sink = acc;
}

template <int MODE> static void run(const char* name, u32x4* a, u32x4* b, u32x4* y, long n, double bytes_per_elem) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int blocks_per_cu = 4; blocks_per_cu <= 16; blocks_per_cu *= 2) {
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            (void)hipEventRecord(e0);
            k<MODE, 4><<<256 * blocks_per_cu, 256>>>(a, b, y, n, y);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < best) best = ms;
        }
        printf("%-28s %2d blocks/CU: %7.1f us  %.2f TB/s\n", name, blocks_per_cu, best * 1e3, n * bytes_per_elem / best / 1e9);
    }
}

int main() {
    const long n = 64L << 20;                 // 64 Mi x 16 B = 1 GiB per stream (4x the 256 MiB Infinity Cache)
    u32x4 *a, *b, *y;
    (void)hipMalloc(&a, n * 16); (void)hipMalloc(&b, n * 16); (void)hipMalloc(&y, n * 16);
    (void)hipMemset(a, 1, n * 16); (void)hipMemset(b, 2, n * 16);
    run<0>("read", a, b, y, n, 16);
    run<1>("write", a, b, y, n, 16);
    run<2>("copy (1 read : 1 write)", a, b, y, n, 32);
    run<3>("2 reads : 1 write", a, b, y, n, 48);
    return 0;
}

// Dev micro-benchmark: what one workgroup barrier costs on gfx950 by workgroup size, alone and with MFMA work between barriers.
//   hipcc -O3 --offload-arch=gfx950 barrier_cost.cpp -o barrier_cost
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

template <int NMFMA, bool HALF>     // HALF: only the first half of the waves issue MFMAs (ping-pong halves), the rest just join the barrier
__global__ void k(float* out, int iters, long long* cycles) {
    f32x4 acc[4] = {};
    s16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {8, 7, 6, 5, 4, 3, 2, 1};
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (!HALF || ((wave < nw / 2) == ((it & 1) == 0)) ) {
#pragma unroll
            for (int m = 0; m < NMFMA; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m & 3], 0, 0, 0);
        }
        __builtin_amdgcn_s_barrier();
    }
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0];
}
template <int NMFMA, bool HALF> static void run(const char* name, int threads) {
    float* out; long long* cyc; long long h = 0;
    (void)hipMalloc(&out, 256 * 1024 * 4); (void)hipMalloc(&cyc, 8);
    const int iters = 2000;
    k<NMFMA, HALF><<<256, threads>>>(out, iters, cyc);
    k<NMFMA, HALF><<<256, threads>>>(out, iters, cyc);
    (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-60s %4d threads: %7.1f cycles per iteration (s_memtime-like counter units)\n", name, threads, (double)h / iters);
    (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
    for (int th : {256, 512, 768, 1024}) run<0, false>("barrier only", th);
    for (int th : {256, 512, 768}) run<28, false>("28 MFMA (16x16x32 bf16) per wave + barrier", th);
    for (int th : {512, 768}) run<28, true>("28 MFMA in alternating halves of the waves + barrier", th);
    for (int th : {512, 768}) run<14, true>("14 MFMA in alternating halves + barrier", th);
    return 0;
}

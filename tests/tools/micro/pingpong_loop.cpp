// Dev micro-benchmark: the compute side of d3q_kernel's K-step in isolation - per interval 18 ds_read_b128 fragment reads of one group
// of four waves beside 28 MFMAs (16x16x32 bf16) of the other group, barrier, roles swapped (PING-PONG), against all eight waves doing
// reads -> MFMAs -> barrier (SAME). 12 waves per block as in the kernel (the last four only join the barriers).
//   hipcc -O3 --offload-arch=gfx950 pingpong_loop.cpp -o pingpong_loop
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

template <bool PINGPONG, int NBAR>
__global__ __launch_bounds__(768, 3) void k(float* out, int iters, long long* cycles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fq = lane >> 4;
    for (int i = threadIdx.x; i < 140 * 1024 / 16; i += 768) reinterpret_cast<f32x4*>(smem)[i] = f32x4{1.f, 2.f, 3.f, 4.f};
    __syncthreads();
    f32x4 acc[2][7] = {};
    s16x8 a[2][2], b[2][7];
    const char* abase = smem + (wave & 3) * 4096 + fr * 128 + ((fq ^ (fr & 7)) << 4);
    const char* bbase = smem + 49152 + (wave >> 2) * 14336 + fr * 128 + ((fq ^ (fr & 7)) << 4);
    auto reads = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int i = 0; i < 2; ++i) a[u][i] = *reinterpret_cast<const s16x8*>(abase + slot * 16384 + i * 2048 + u * 64);
#pragma unroll
            for (int j = 0; j < 7; ++j) b[u][j] = *reinterpret_cast<const s16x8*>(bbase + (slot & 1) * 29696 + j * 2048 + u * 64);
        }
    };
    auto mfmas = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int j = 0; j < 7; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u][i], b[u][j], acc[i][j], 0, 0, 0);
    };
    const long long t0 = __builtin_readcyclecounter();
    if (wave >= 8) {
        for (int it = 0; it < iters * NBAR; ++it) __builtin_amdgcn_s_barrier();
    } else if (PINGPONG) {
        const int grp = wave >> 2;
        for (int it = 0; it < iters; ++it) {                 // one K-step: two intervals
            if (grp == 0) { reads(it % 3); __builtin_amdgcn_s_waitcnt(0xC07F); } else if (it > 0) mfmas();
            __builtin_amdgcn_s_barrier();
            if (grp == 0) mfmas(); else { reads(it % 3); __builtin_amdgcn_s_waitcnt(0xC07F); }
            __builtin_amdgcn_s_barrier();
        }
    } else {
        for (int it = 0; it < iters; ++it) {
            reads(it % 3);
            mfmas();
            for (int q = 0; q < NBAR; ++q) __builtin_amdgcn_s_barrier();
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
    float r = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 7; ++j) r += acc[i][j][0];
    out[blockIdx.x * 768 + threadIdx.x] = r;
}
template <bool PP, int NBAR> static void run(const char* name) {
    float* out; long long* cyc; long long h = 0;
    (void)hipMalloc(&out, 256 * 768 * 4); (void)hipMalloc(&cyc, 8);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<PP, NBAR>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    const int iters = 1000;
    for (int r = 0; r < 2; ++r) k<PP, NBAR><<<256, 768, 140 * 1024>>>(out, iters, cyc);
    (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-70s %7.1f cycles per K-step (28 MFMA per wave, 8 compute waves; pure MFMA time 896)\n", name, (double)h / iters);
    (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
    run<true, 2>("ping-pong groups, 2 barriers per K-step (d3q_body)");
    run<false, 2>("all waves reads -> MFMAs, 2 barriers per K-step");
    run<false, 1>("all waves reads -> MFMAs, 1 barrier per K-step");
    return 0;
}

// Dev micro-benchmark: the compute side of a d3q K-step as TWO INDEPENDENT 6-wave blocks per CU (4 compute waves + 2 waves that only
// join the barrier, as loader waves would) - every compute wave: 18 ds_read_b128 fragment reads -> 28 MFMAs (16x16x32 bf16), one
// barrier per K-step - against pingpong_loop.cpp's one 12-wave block per CU (two compute groups alternating, two barriers per K-step).
// Per SIMD and K-step-pair both forms owe the matrix pipe 2 x 448 = 896 cycles... here: a block's K-step = 28 MFMAs per wave = 448
// pipe cycles, two blocks per CU -> 896 pipe cycles per (K-step of block X + K-step of block Y).
//   hipcc -O3 --offload-arch=gfx950 twoblock_loop.cpp -o twoblock_loop
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

template <int KH>   // KH: K-halves per K-step (2: 18 reads + 28 MFMAs ... per half: 9 reads + 14 MFMAs)
__global__ __launch_bounds__(384, 3) void k(float* out, int iters, long long* cycles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fq = lane >> 4;
    for (int i = threadIdx.x; i < 72 * 1024 / 16; i += 384) reinterpret_cast<f32x4*>(smem)[i] = f32x4{1.f, 2.f, 3.f, 4.f};
    __syncthreads();
    f32x4 acc[2][7] = {};
    s16x8 a[2][2], b[2][7];
    const char* abase = smem + (wave & 1) * 4096 + fr * 128 + ((fq ^ (fr & 7)) << 4);
    const char* bbase = smem + 24576 + ((wave >> 1) & 1) * 14336 + fr * 128 + ((fq ^ (fr & 7)) << 4);
    auto reads = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int i = 0; i < 2; ++i) a[u][i] = *reinterpret_cast<const s16x8*>(abase + slot * 8192 + i * 2048 + u * 64);
#pragma unroll
            for (int j = 0; j < 7; ++j) b[u][j] = *reinterpret_cast<const s16x8*>(bbase + (slot & 1) * 1024 + j * 2048 + u * 64);
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
    if (wave >= 4) {
        for (int it = 0; it < iters; ++it) __builtin_amdgcn_s_barrier();
    } else {
        for (int it = 0; it < iters; ++it) {
            reads(it % 3);
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_s_setprio(1);
            mfmas();
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_s_barrier();
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
    float r = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 7; ++j) r += acc[i][j][0];
    out[blockIdx.x * 384 + threadIdx.x] = r;
}
int main() {
    float* out; long long* cyc; long long h = 0;
    (void)hipMalloc(&out, 512 * 384 * 4); (void)hipMalloc(&cyc, 8);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
    int nb = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(k<2>), 384, 72 * 1024);
    const int iters = 1000;
    for (int grid : {256, 512}) {
        for (int r = 0; r < 2; ++r) k<2><<<grid, 384, 72 * 1024>>>(out, iters, cyc);
        (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("grid %d (%d block(s) per CU; occupancy query %d): %7.1f cycles per block K-step (28 MFMA per wave = 448 pipe cycles per block; "
               "with two blocks per CU the pipe owes 896 per pair)\n", grid, grid / 256, nb, (double)h / iters);
    }
    return 0;
}

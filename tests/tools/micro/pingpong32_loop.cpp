// Dev micro-benchmark: pingpong_loop.cpp with v_mfma_f32_32x32x16_bf16 - wave tile 32 channels x 128 pixels (1 x 4 blocks of
// 32 x 32), per K-step (K = 64 = 4 slices of 16) 4 + 16 ds_read_b128 and 16 MFMAs of 32 cycles = 512 pipe cycles per wave, 1 024
// per SIMD and K-step; against 28 MFMAs of 16 cycles (wave tile 32 x 112, 896 per SIMD) in pingpong_loop.cpp.
//   hipcc -O3 --offload-arch=gfx950 pingpong32_loop.cpp -o pingpong32_loop
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

template <bool PINGPONG, int NBAR>
__global__ __launch_bounds__(768, 3) void k(float* out, int iters, long long* cycles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r32 = lane & 31, half = lane >> 5;
    for (int i = threadIdx.x; i < 150 * 1024 / 16; i += 768) reinterpret_cast<f32x4*>(smem)[i] = f32x4{1.f, 2.f, 3.f, 4.f};
    __syncthreads();
    f32x16 acc[4] = {};
    s16x8 a[4], b[4][4];          // [k16 slice], [k16 slice][pixel block]
    // row r32 of a 32-row block, 16-byte chunk (2 * slice + half) of its 128-byte row; XOR swizzle on the row as in the kernels
    const char* abase = smem + (wave & 3) * 4096 + r32 * 128;
    const char* bbase = smem + 49152 + ((wave >> 2) & 1) * 16384 + r32 * 128;
    auto reads = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int ch = ((2 * s + half) ^ (r32 & 7)) << 4;
            a[s] = *reinterpret_cast<const s16x8*>(abase + slot * 16384 + ch);
#pragma unroll
            for (int j = 0; j < 4; ++j) b[s][j] = *reinterpret_cast<const s16x8*>(bbase + (slot & 1) * 33792 + j * 4096 + ch);
        }
    };
    auto mfmas = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], b[s][j], acc[j], 0, 0, 0);
    };
    const long long t0 = __builtin_readcyclecounter();
    if (wave >= 8) {
        for (int it = 0; it < iters * NBAR; ++it) __builtin_amdgcn_s_barrier();
    } else if (PINGPONG) {
        const int grp = wave >> 2;
        for (int it = 0; it < iters; ++it) {
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
    for (int j = 0; j < 4; ++j) r += acc[j][0] + acc[j][7];
    out[blockIdx.x * 768 + threadIdx.x] = r;
}
template <bool PP, int NBAR> static void run(const char* name) {
    float* out; long long* cyc; long long h = 0;
    (void)hipMalloc(&out, 256 * 768 * 4); (void)hipMalloc(&cyc, 8);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<PP, NBAR>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    const int iters = 1000;
    for (int r = 0; r < 2; ++r) k<PP, NBAR><<<256, 768, 150 * 1024>>>(out, iters, cyc);
    (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-62s %7.1f cycles per K-step (16 MFMA 32x32x16 per wave; pure MFMA time 1024) -> %.0f %% of the pipe\n", name, (double)h / iters,
           100.0 * 1024 / ((double)h / iters));
    (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
    run<true, 2>("32x32x16: ping-pong groups, 2 barriers per K-step");
    run<false, 2>("32x32x16: all waves reads -> MFMAs, 2 barriers per K-step");
    run<false, 1>("32x32x16: all waves reads -> MFMAs, 1 barrier per K-step");
    return 0;
}

// Dev micro-benchmark, companion of mfma_loop_real.cpp: the same loop cut into HALF K-steps - two 16 KB LDS stages per block
// (64-byte operand rows), 16 MFMAs and 4 one-KB DMA pieces per wave per step, one barrier per step - so that 4 blocks
// (16 waves) fit a CU instead of 2. Same DMA bytes per MFMA, twice the barriers, twice the waves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ __launch_bounds__(256, 4) void k(const s16x8* __restrict__ src, float* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];          // 2 stages x 16 KB
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<s16x8*>(src), 0, 4 << 20, 0x00020000);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 32768 / 16; i += 256) reinterpret_cast<s16x8*>(smem)[i] = src[i];
    __syncthreads();
    f32x4 acc[4][4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        __syncthreads();
        char* nxt = smem + ((it + 1) & 1) * 16384;
#pragma unroll
        for (int d = 0; d < 4; ++d)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(nxt + (wave * 4 + d) * 1024), 16,
                                                     (((uint32_t)(it * 4 + d) * 40503u + blockIdx.x * 9973u + wave * 613u) & 4095u) * 1024u + lane * 16u,
                                                     0, 0, 0);
        const s16x8* cur = reinterpret_cast<const s16x8*>(smem + (it & 1) * 16384);
        s16x8 a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[i] = cur[i * 64 + lane];
            b[i] = cur[(4 + i) * 64 + lane + 512];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * 256 + tid] = s;
}

int main() {
    std::vector<short> h((4 << 20) / 2);
    for (auto& v : h) v = (short)(0x3c00 + (rand() & 0x3ff) - ((rand() & 1) ? 0x8000 : 0));
    s16x8* src; float* out;
    if (hipMalloc(&src, 4 << 20) != hipSuccess || hipMalloc(&out, 1024 * 256 * 4) != hipSuccess) return 1;
    (void)hipMemcpy(src, h.data(), 4 << 20, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000;
    for (int blocks = 256; blocks <= 1024; blocks *= 2) {
        float best = 1e9;
        for (int rep = 0; rep < 4; ++rep) {
            (void)hipEventRecord(e0);
            k<<<blocks, 256, 32768>>>(src, out, iters);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < best) best = ms;
        }
        const double flop = (double)blocks * 4 * iters * 16 * (2.0 * 16 * 16 * 32);
        printf("half K-steps, %d block(s) per CU: %6.1f ms  %5.0f TFLOP/s\n", blocks / 256, best, flop / best / 1e9);
    }
    return 0;
}

// Dev micro-benchmark: the K loop of igemm_conv.hpp at its real resource footprint - 4 waves per block, two 32 KB LDS stages
// (64 KB per block -> two blocks per CU), every wave DMAs 8 one-KB pieces of the NEXT stage from an L2-resident 4 MB source
// while the 32 MFMAs of the step read all operands of the CURRENT stage with ds_read_b128; one __syncthreads per step.
// No address generation beyond that, no epilogue, no tile schedule: what is left is the hardware's rate for this loop shape.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int NDMA>
__global__ __launch_bounds__(256, 2) void k(const s16x8* __restrict__ src, float* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];          // 2 stages x 32 KB
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<s16x8*>(src), 0, 4 << 20, 0x00020000);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 65536 / 16; i += 256) reinterpret_cast<s16x8*>(smem)[i] = src[i];
    __syncthreads();
    f32x4 acc[4][4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        __syncthreads();
        char* nxt = smem + ((it + 1) & 1) * 32768;
#pragma unroll
        for (int d = 0; d < NDMA; ++d)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(nxt + (wave * 8 + d) * 1024), 16,
                                                     (((uint32_t)(it * 8 + d) * 40503u + blockIdx.x * 9973u + wave * 613u) & 4095u) * 1024u + lane * 16u,
                                                     0, 0, 0);
        const s16x8* cur = reinterpret_cast<const s16x8*>(smem + (it & 1) * 32768);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            s16x8 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = cur[(ks * 8 + i) * 64 + lane];                     // 16 fragments of 1 KB = 16 KB of the 32 KB stage
                b[i] = cur[(ks * 8 + 4 + i) * 64 + lane + 1024];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * 256 + tid] = s;
}

int main() {
    std::vector<short> h((4 << 20) / 2);
    for (auto& v : h) v = (short)(0x3c00 + (rand() & 0x3ff) - ((rand() & 1) ? 0x8000 : 0));     // bf16 around +-1
    s16x8* src; float* out;
    if (hipMalloc(&src, 4 << 20) != hipSuccess || hipMalloc(&out, 512 * 256 * 4) != hipSuccess) return 1;
    (void)hipMemcpy(src, h.data(), 4 << 20, hipMemcpyHostToDevice);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 10000;
    for (int nd = 8; nd >= 4; nd -= 4)
        for (int blocks = 256; blocks <= 512; blocks *= 2) {
            float best = 1e9;
            for (int rep = 0; rep < 4; ++rep) {
                (void)hipEventRecord(e0);
                if (nd == 8) k<8><<<blocks, 256, 65536>>>(src, out, iters); else k<4><<<blocks, 256, 65536>>>(src, out, iters);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0 && ms < best) best = ms;
            }
            const double flop = (double)blocks * 4 * iters * 2 * 16 * (2.0 * 16 * 16 * 32);
            printf("%d pieces per wave per step, %d block(s) per CU: %6.1f ms  %5.0f TFLOP/s\n", nd, blocks / 256, best, flop / best / 1e9);
        }
    return 0;
}

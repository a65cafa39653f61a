// Dev micro-benchmark: the MFMA rate this MI355X sustains (a) on register operands and (b) when every operand of a 64x64
// wave tile (4 A + 4 B fragments per 16 v_mfma_f32_16x16x32_bf16) is re-read from LDS with ds_read_b128 - the inner loop
// shape of igemm_conv.hpp without any global traffic, barriers or epilogue. Random data (the chip clocks differently on
// zeros). hipcc -O3 --offload-arch=gfx950 mfma_lds_ceiling.cpp -o mfma_lds_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// MODE bit 0: operands re-read from LDS; bit 1: one __syncthreads per 32 MFMAs (the K-step barrier of the conv kernels);
// bit 2: NDMA LDS-DMA pieces (1 KB each, L2-resident source) issued per 32 MFMAs into a third LDS stage nobody reads.
template <int MODE, int NDMA, int TPB = 256>
__global__ __launch_bounds__(TPB, 512 / TPB) void k(const s16x8* __restrict__ src, float* __restrict__ out, int iters) {
    constexpr bool FROM_LDS = (MODE & 1) != 0;
    __shared__ __attribute__((aligned(16))) char dma_stage[2][NDMA > 0 ? NDMA * (TPB / 64) * 1024 : 16];
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<s16x8*>(src), 0, (MODE & 32) ? (4 << 20) : 2 * 8 * 64 * 2 * 16, 0x00020000);
    __shared__ __attribute__((aligned(16))) s16x8 lds[2][8 * 64 * 2];      // 2 stages x (8 fragments x 64 lanes) x 2 K-steps
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 2 * 8 * 64 * 2; i += TPB) (&lds[0][0])[i] = src[i];
    __syncthreads();
    f32x4 acc[4][4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    s16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = lds[0][i * 64 + lane]; b[i] = lds[0][(4 + i) * 64 + lane]; }
    const int wave = tid >> 6;
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    u32x4 apre[8];
    if (MODE & 16) {
#pragma unroll
        for (int f = 0; f < 8; ++f) apre[f] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (uint32_t)((f * 64 + lane) * 16), 0, 0);
    }
    for (int it = 0; it < iters; ++it) {
        if (MODE & 2) __syncthreads();          // as in the conv kernels: barrier (waits for the DMA issued one step ago) ...
        if (MODE & 4) {                           // ... then the next stage's DMA, then this stage's MFMAs
#pragma unroll
            for (int d = 0; d < NDMA; ++d)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(dma_stage[it & 1] + (wave * NDMA + d) * 1024),
                                                         16, (MODE & 32) ? (((uint32_t)(it * NDMA + d) * 40503u + blockIdx.x * 9973u + wave * 613u) & 4095u) * 1024u + lane * 16u
                                                                         : (uint32_t)(((it * 7 + d) & 31) * 1024 + lane * 16), 0, 0, 0);
        }
        u32x4 acur[8];
        if (MODE & 16) {
#pragma unroll
            for (int f = 0; f < 8; ++f) acur[f] = apre[f];
#pragma unroll
            for (int f = 0; f < 8; ++f)          // next K-step's A fragments: 16 contiguous bytes per lane, L2-resident weights
                apre[f] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (uint32_t)(((((it + 1) * 5 + f) & 15) * 64 + lane) * 16), 0, 0);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (FROM_LDS) {
                const s16x8* st = &lds[it & 1][ks * 8 * 64];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (MODE & 16) a[i] = __builtin_bit_cast(s16x8, acur[ks * 4 + i]); else a[i] = st[i * 64 + lane];
                    b[i] = st[(4 + i) * 64 + lane];
                }
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
    out[blockIdx.x * TPB + tid] = s;
}

int main() {
    const int n = 2 * 8 * 64 * 2;
    std::vector<short> h(n * 8);
    for (auto& v : h) v = (short)(0x3c00 + (rand() & 0x3ff) - ((rand() & 1) ? 0x8000 : 0));     // bf16 around +-1
    s16x8* src; float* out;
    if (hipMalloc(&src, 4 << 20) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(src, 0x3c, 4 << 20); (void)hipMalloc(&out, 512 * 512 * 4);
    (void)hipMemcpy(src, h.data(), n * 16, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 10000;
    auto run8 = [&](const char* name, void (*fn)(const s16x8*, float*, int)) {     // one 8-wave block per CU
        float best = 1e9;
        for (int rep = 0; rep < 4; ++rep) {
            (void)hipEventRecord(e0);
            fn<<<256, 512>>>(src, out, iters);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < best) best = ms;
        }
        const double flop = 256.0 * 8 * iters * 2 * 16 * (2.0 * 16 * 16 * 32);
        printf("%-58s 8-wave block  : %6.1f ms  %5.0f TFLOP/s\n", name, best, flop / best / 1e9);
    };
    auto run = [&](const char* name, void (*fn)(const s16x8*, float*, int)) {
        for (int blocks = 256; blocks <= 512; blocks *= 2) {
            float best = 1e9;
            for (int rep = 0; rep < 4; ++rep) {
                (void)hipEventRecord(e0);
                fn<<<blocks, 256>>>(src, out, iters);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0 && ms < best) best = ms;
            }
            const double flop = (double)blocks * 4 * iters * 2 * 16 * (2.0 * 16 * 16 * 32);
            printf("%-58s %d waves/SIMD: %6.1f ms  %5.0f TFLOP/s\n", name, blocks / 256, best, flop / best / 1e9);
        }
    };
    run("register operands", k<0, 0>);
    run("operands from LDS (4 A + 4 B per 16 MFMA)", k<1, 0>);
    run("  + barrier per 32 MFMA", k<3, 0>);
    run("  + 4 LDS-DMA pieces per wave per 32 MFMA", k<5, 4>);
    run("  + 8 LDS-DMA pieces per wave per 32 MFMA", k<5, 8>);
    run("  + barrier + 4 pieces", k<7, 4>);
    run("  + barrier + 8 pieces (generic 128x128 tile)", k<7, 8>);
    run8("  + barrier + 2 pieces", k<7, 2, 512>);
    run8("  + barrier + 4 pieces", k<7, 4, 512>);
    run8("  + barrier + 8 pieces", k<7, 8, 512>);
    run("  + barrier + 4 pieces, source = 4 MB in L2 (not L1)", k<39, 4>);
    run8("  + barrier + 4 pieces, source = 4 MB in L2 (not L1)", k<39, 4, 512>);
    run8("  + barrier + 8 pieces, source = 4 MB in L2 (not L1)", k<39, 8, 512>);
    run("A: 8 global loads -> regs; B: LDS + barrier + 4 pieces", k<23, 4>);
    run("A: 8 global loads -> regs; B: LDS + barrier + 8 pieces", k<23, 8>);
    return 0;
}

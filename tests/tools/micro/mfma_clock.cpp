// Dev micro-benchmark (GPU box): what a BARE v_mfma_f32_16x16x32_bf16 loop sustains on this chip, and at which clock - the ceiling every
// dense-convolution kernel of this repository is priced against. 256 blocks x 4 waves (one per SIMD, 512-register budget), operands in
// registers, 52 independent accumulators per wave (d3i_kernel's wave tile), no memory traffic inside the loop; optionally one ds_read_b128
// per four MFMAs (d3i's ratio). Random bf16 operands against all-zero operands: the chip holds its clock down under load
// (MI355X_MICROARCH.md, "DVFS give-back"), so the 2.5 PFLOP/s headline (2.4 GHz x 1 024 FLOP/clk/SIMD) is not what random data reaches.
// In-kernel clock = delta s_memtime / delta s_memrealtime x 100 MHz (median over waves), after ~1 s of back-to-back launches.
//   hipcc -O3 --offload-arch=gfx950 mfma_clock.cpp -o /tmp/mfma_clock && /tmp/mfma_clock
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <bool LDSR>
__global__ __launch_bounds__(256, 1) void k(const s16x8* __restrict__ src, float* __restrict__ out, unsigned long long* __restrict__ clk, int iters) {
    __shared__ __attribute__((aligned(16))) s16x8 lds[13 * 64 * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    s16x8 a[4], b[13];
    for (int i = 0; i < 4; ++i) a[i] = src[(blockIdx.x * 17 + i) * 64 + lane];
    for (int i = 0; i < 13; ++i) b[i] = src[(4096 + wave * 13 + i) * 64 + lane];
    for (int i = 0; i < 13; ++i) lds[(wave * 13 + i) * 64 + lane] = b[i];
    __syncthreads();
    f32x4 acc[13][4];
    for (int i = 0; i < 13; ++i)
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 13; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j)       // (inline asm: through the builtin the register allocator copies all 208 accumulators around the back edge)
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(a[j]), "v"(b[i]));
            if (LDSR) b[i] = lds[(wave * 13 + i) * 64 + ((lane + (it & 1)) & 63)];       // refilled for the next pass, 48 MFMAs ahead of its use
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");          // (the asm MFMAs' results, before the compiler's code reads them)
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 13; ++i)
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * 256 + tid] = s;
    if (lane == 0) { clk[(blockIdx.x * 4 + wave) * 2] = c1 - c0; clk[(blockIdx.x * 4 + wave) * 2 + 1] = r1 - r0; }
}

int main() {
    const size_t n = (size_t)(4096 + 64) * 64 * 8;
    std::vector<short> h(n);
    s16x8* src; float* out; unsigned long long* clk;
    if (hipMalloc(&src, n * 2) != hipSuccess || hipMalloc(&out, 256 * 256 * 4) != hipSuccess || hipMalloc(&clk, 256 * 4 * 16) != hipSuccess) return 1;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 4000;                      // 4000 x 52 MFMAs = 3.3 M cycles per launch (~2 ms)
    for (int data = 0; data < 3; ++data) {       // 0: random bf16 around +-1 (activations x weights), 1: random bit patterns of finite bf16, 2: zeros
        for (auto& v : h) v = data == 0 ? (short)(0x3c00 + (rand() & 0x3ff) - ((rand() & 1) ? 0x8000 : 0))
                              : data == 1 ? (short)((rand() & 0x7fff) % 0x7f00 | ((rand() & 1) << 15)) : (short)0;
        (void)hipMemcpy(src, h.data(), n * 2, hipMemcpyHostToDevice);
        for (int ldsr = 0; ldsr < 2; ++ldsr)
            for (int blocks = 64; blocks <= 256; blocks *= 4) {
                auto launch = [&]() { if (ldsr) k<true><<<blocks, 256>>>(src, out, clk, iters); else k<false><<<blocks, 256>>>(src, out, clk, iters); };
                for (int w = 0; w < 400; ++w) launch();                     // ~1 s of load before the measurement
                (void)hipEventRecord(e0);
                for (int w = 0; w < 50; ++w) launch();
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                std::vector<unsigned long long> c((size_t)blocks * 8);
                (void)hipMemcpy(c.data(), clk, c.size() * 8, hipMemcpyDeviceToHost);
                std::vector<double> ghz, cyc;
                for (int i = 0; i < blocks * 4; ++i) { ghz.push_back((double)c[2 * i] / (double)c[2 * i + 1] * 0.1); cyc.push_back((double)c[2 * i] / (iters * 52.0)); }
                std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
                const double flop = (double)blocks * 4 * iters * 52 * (2.0 * 16 * 16 * 32) * 50;
                printf("%-22s %-26s %3d blocks: %7.0f TFLOP/s   clock %.2f GHz (min %.2f max %.2f)   %.2f cycles per MFMA\n",
                       data == 0 ? "random bf16 around +-1" : data == 1 ? "random finite bf16" : "zeros", ldsr ? "1 ds_read_b128 per 4 MFMAs" : "registers only", blocks,
                       flop / ms / 1e9, ghz[ghz.size() / 2], ghz.front(), ghz.back(), cyc[cyc.size() / 2]);
                fflush(stdout);
            }
    }
    return 0;
}

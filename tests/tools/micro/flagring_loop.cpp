// Dev micro-benchmark: the compute side of a d3q K-step with NO workgroup barrier in the loop - eight compute waves run free
// (poll the slot's FULL counter in LDS -> 18 ds_read_b128 -> bump the slot's FREE counter -> 28 MFMAs), four "loader" waves only
// play the ring protocol (wait for FREE of the slot they refill, bump FULL): what a flag-synchronised ring costs beside
// pingpong_loop.cpp's barrier-synchronised groups (1 134 cycles per K-step) and twoblock_loop.cpp (880 per pair).
//   hipcc -O3 --offload-arch=gfx950 flagring_loop.cpp -o flagring_loop
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

__device__ __forceinline__ bool wait_ge(volatile unsigned* ctr, unsigned target) {
    for (int spin = 0; spin < (1 << 20); ++spin) {
        if (*ctr >= target) return true;
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}

template <int EPI>     // EPI: every 36th K-step a wave spends ~EPI cycles in a VALU-only "epilogue" (0 = none)
__global__ __launch_bounds__(768, 3) void k(float* out, int iters, long long* cycles, int* err) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ unsigned full[3], freec[3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fq = lane >> 4;
    for (int i = threadIdx.x; i < 140 * 1024 / 16; i += 768) reinterpret_cast<f32x4*>(smem)[i] = f32x4{1.f, 2.f, 3.f, 4.f};
    if (threadIdx.x < 3) { full[threadIdx.x] = 0; freec[threadIdx.x] = 0; }
    __syncthreads();
    f32x4 acc[2][7] = {};
    s16x8 a[2][2], b[2][7];
    const char* abase = smem + (wave & 3) * 4096 + fr * 128 + ((fq ^ (fr & 7)) << 4);
    const char* bbase = smem + 49152 + ((wave >> 2) & 1) * 14336 + fr * 128 + ((fq ^ (fr & 7)) << 4);
    const long long t0 = __builtin_readcyclecounter();
    bool ok = true;
    if (wave >= 8) {
        // loader: K-step s fills slot s % 3 (last read by K-step s - 3): wait until all 8 compute waves released that use
        for (int s = 0; s < iters && ok; ++s) {
            const int slot = s % 3, use = s / 3;
            if (use > 0) ok = wait_ge(&freec[slot], 8u * use);
            __builtin_amdgcn_s_sleep(8);                               // (stands for issuing + landing of the pieces)
            if (lane == 0) atomicAdd(&full[slot], 1u);
        }
    } else {
        float junk = 0.f;
        for (int s = 0; s < iters && ok; ++s) {
            const int slot = s % 3, use = s / 3;
            ok = wait_ge(&full[slot], 4u * (use + 1));
#pragma unroll
            for (int u = 0; u < 2; ++u) {
#pragma unroll
                for (int i = 0; i < 2; ++i) a[u][i] = *reinterpret_cast<const s16x8*>(abase + slot * 16384 + i * 2048 + u * 64);
#pragma unroll
                for (int j = 0; j < 7; ++j) b[u][j] = *reinterpret_cast<const s16x8*>(bbase + (slot & 1) * 29696 + j * 2048 + u * 64);
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            if (lane == 0) atomicAdd(&freec[slot], 1u);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < 7; ++j)
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u][i], b[u][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            if (EPI > 0 && s % 36 == 35) {
                for (int e = 0; e < EPI / 8; ++e) junk = junk * 1.0001f + acc[e & 1][e % 7][0];   // dependent VALU chain ~ 8 cycles each
            }
        }
        acc[0][0][0] += junk;
    }
    const long long t1 = __builtin_readcyclecounter();
    if (!ok) *err = 1;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
    float r = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 7; ++j) r += acc[i][j][0];
    out[blockIdx.x * 768 + threadIdx.x] = r;
}
template <int EPI> static void run(const char* name) {
    float* out; long long* cyc; int* err; long long h = 0; int he = 0;
    (void)hipMalloc(&out, 256 * 768 * 4); (void)hipMalloc(&cyc, 8); (void)hipMalloc(&err, 4); (void)hipMemset(err, 0, 4);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    const int iters = 1008;
    for (int r = 0; r < 2; ++r) k<EPI><<<256, 768, 140 * 1024>>>(out, iters, cyc, err);
    (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&he, err, 4, hipMemcpyDeviceToHost);
    printf("%-58s %7.1f cycles per K-step (pure MFMA time 896)%s\n", name, (double)h / iters, he ? "  TIMEOUT" : "");
}
int main() {
    run<0>("flag ring, free-running compute waves");
    run<3600>("... + a 3600-cycle VALU epilogue per wave every 36 K-steps");
    return 0;
}

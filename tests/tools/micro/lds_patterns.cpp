// Dev micro-benchmark: LDS cycles per instruction of the access patterns of mbw.hpp (64-byte rows, slot XOR (row >> 1) & 2), one wave
// per CU so that only bank conflicts (not other waves) stretch an instruction. Each pattern issues 16 independent instructions per
// iteration; the linear pattern (lane l -> byte 16 l) is the conflict-free reference.
//   hipcc -O3 --offload-arch=gfx950 lds_patterns.cpp -o lds_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

__device__ __forceinline__ int swz(int row) { return (row >> 1) & 2; }

// kind: 0 linear b128 read, 1 fragment read (row fr, slot fq ^ swz), 2 S2 B read (1 x 16 block, stride 1, window pitch 18), 3 S2 B read of a
// 2 x 8 block (stride 1, window pitch 10, rows R = 4 apart), 4 D write b64, 5 linear b64 write, 6 u16 weight read, 7 E write b128,
// 8 S2 B read 2 x 8 stride 2 (window pitch 18, even / odd column halves)
template <int KIND> __global__ __launch_bounds__(64) void k(unsigned* out, int iters, long long* cyc) {
    __shared__ __attribute__((aligned(16))) char smem[32768];
    const int lane = threadIdx.x, fr = lane & 15, fq = lane >> 4;
    for (int i = lane; i < 32768 / 4; i += 64) reinterpret_cast<unsigned*>(smem)[i] = i;
    __syncthreads();
    int off[16];
    for (int t = 0; t < 16; ++t) {
        const int u = t & 3, j = (t >> 2) % 5, g = (t >> 2) & 1;
        if (KIND == 0 || KIND == 5) off[t] = lane * (KIND == 0 ? 16 : 8) + t * 1024;
        else if (KIND == 1) off[t] = (16 * t + fr) * 64 + ((fq ^ swz(fr)) << 4);
        else if (KIND == 2) {
            const int tap = min(2 * j + (fq >> 1), 8);
            const int sidx = (u + tap / 3) * 18 + fr + tap % 3;
            off[t] = (sidx * 64 + (((fq & 1) ^ swz(sidx)) << 4)) ^ (g ? 32 : 0);
        } else if (KIND == 3) {
            const int tap = min(2 * j + (fq >> 1), 8);
            const int pr = fr / 8, pc = fr % 8;
            const int sidx = (u + 4 * pr + tap / 3) * 10 + pc + tap % 3;
            off[t] = (sidx * 64 + (((fq & 1) ^ swz(sidx)) << 4)) ^ (g ? 32 : 0);
        } else if (KIND == 8) {
            const int tap = min(2 * j + (fq >> 1), 8);
            const int pr = fr / 8, pc = fr % 8;
            const int wr = (u + 2 * pr) * 2 + tap / 3, wcol = pc * 2 + tap % 3;
            const int sidx = wr * 18 + (wcol & 1) * 9 + (wcol >> 1);
            off[t] = (sidx * 64 + (((fq & 1) ^ swz(sidx)) << 4)) ^ (g ? 32 : 0);
        } else if (KIND == 4) off[t] = (16 * u + fr) * 64 + ((((2 * g + (fq >> 1)) ^ swz(fr))) << 4) + 8 * (fq & 1) + (t >> 3) * 4096;
        else if (KIND == 6) off[t] = ((2 * j + (fq >> 1)) * 144 + 32 * u + 16 * g + fr) * 2;
        else if (KIND == 7) off[t] = (16 * t + fr) * 64 + ((fq ^ swz(fr)) << 4);
        else if (KIND == 9) off[t] = lane * 16 + t * 1024;                                        // linear b128 write
        else if (KIND == 10 || KIND == 11 || KIND == 12) {                                        // slot-major 16-pixel blocks: addr(q, s)
            const int tap = min(2 * j + (fq >> 1), 8);
            int sidx;
            if (KIND == 10) sidx = (u + tap / 3) * 18 + fr + tap % 3;
            else if (KIND == 11) { const int pr = fr / 8, pc = fr % 8; sidx = (u + 4 * pr + tap / 3) * 10 + pc + tap % 3; }
            else { const int pr = fr / 8, pc = fr % 8; const int wr = (u + 2 * pr) * 2 + tap / 3, wcol = pc * 2 + tap % 3; sidx = wr * 18 + (wcol & 1) * 9 + (wcol >> 1); }
            const int sl = (fq & 1) + 2 * g;
            off[t] = (sidx >> 4) * 1024 + sl * 256 + (sidx & 15) * 16;
        } else if (KIND == 13) off[t] = t * 1024 + fq * 256 + fr * 16;                           // fragment read of a slot-major block
        else off[t] = 0;
    }
    unsigned acc = 0;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 4 || KIND == 5) {
#pragma unroll
            for (int t = 0; t < 16; ++t) *reinterpret_cast<u32x2*>(smem + off[t]) = u32x2{acc + t, (unsigned)it};
        } else if (KIND == 7 || KIND == 9) {
#pragma unroll
            for (int t = 0; t < 16; ++t) *reinterpret_cast<u32x4*>(smem + off[t]) = u32x4{acc + t, (unsigned)it, 1u, 2u};
        } else if (KIND == 6) {
#pragma unroll
            for (int t = 0; t < 16; ++t) acc += *reinterpret_cast<const unsigned short*>(smem + off[t]);
        } else {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(smem + off[t]);
                acc += v[0] ^ v[3];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const long long t1 = __builtin_readcyclecounter();
    if (lane == 0 && blockIdx.x == 0) *cyc = t1 - t0;
    out[blockIdx.x * 64 + lane] = acc + reinterpret_cast<unsigned*>(smem)[lane];
}
template <int KIND> static void run(const char* name) {
    unsigned* out; long long* cyc; long long h = 0;
    (void)hipMalloc(&out, 256 * 64 * 4); (void)hipMalloc(&cyc, 8);
    const int iters = 2000;
    for (int r = 0; r < 2; ++r) k<KIND><<<256, 64>>>(out, iters, cyc);
    (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-70s %6.1f cycles per instruction\n", name, (double)h / iters / 16);
    (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
    run<0>("linear ds_read_b128 (reference)");
    run<1>("fragment read: row fr, slot fq ^ swz (weights, D)");
    run<2>("S2 operand read, 1 x 16 block, stride 1");
    run<3>("S2 operand read, 2 x 8 block, stride 1");
    run<8>("S2 operand read, 2 x 8 block, stride 2");
    run<7>("E write ds_write_b128: row fr, slot fq ^ swz");
    run<9>("linear ds_write_b128 (reference)");
    run<10>("slot-major blocks: S2 operand read, 1 x 16 block, stride 1");
    run<11>("slot-major blocks: S2 operand read, 2 x 8 block, stride 1");
    run<12>("slot-major blocks: S2 operand read, 2 x 8 block, stride 2");
    run<13>("slot-major blocks: fragment read (= linear)");
    run<5>("linear ds_write_b64 (reference)");
    run<4>("D write ds_write_b64: row fr, slot (2 g + fq / 2) ^ swz, half fq & 1");
    run<6>("depthwise tap read ds_read_u16");
    return 0;
}

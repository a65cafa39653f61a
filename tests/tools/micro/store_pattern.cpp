// Dev micro-benchmark: does the NHWC epilogue store pattern (16 pixels x 64-byte segments per wave instruction) cost
// HBM write bandwidth compared with full 128-byte lines (8 pixels x 128 bytes)?  hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// each wave writes tiles of 64 pixels x 128 bytes (its 64-channel slice) into rows of `pitch` bytes
template <int PATTERN>
__global__ __launch_bounds__(256) void k(char* y, long ntiles_per_wave, int pitch, long wave_stride_tiles) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const u32x4 v = {1u, 2u, 3u, (unsigned)lane};
    for (long t = 0; t < ntiles_per_wave; ++t) {
        char* base = y + (wave * ntiles_per_wave + t) * 64 * (long)pitch;
        if (PATTERN == 0) {
            const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
            for (int ip = 0; ip < 2; ++ip)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    *reinterpret_cast<u32x4*>(base + (long)(16 * j + fr) * pitch + ip * 64 + fq * 16) = v;
        } else {
            const int px = lane >> 3, ch = lane & 7;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                *reinterpret_cast<u32x4*>(base + (long)(8 * j + px) * pitch + ch * 16) = v;
        }
    }
}

int main() {
    const int pitches[2] = {128, 512};
    for (int pi = 0; pi < 2; ++pi) {
        const int pitch = pitches[pi];
        const long total_rows = 256L * 112 * 112 * 2;       // pixels (0.8 - 3.3 GB written: far beyond the 256 MB cache)
        const long waves = 256 * 8 * 4;                     // 2048 blocks x 4 waves
        const long tiles = total_rows / 64;
        const long tpw = tiles / waves;
        char* y;
        (void)hipMalloc(&y, total_rows * pitch);
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int pat = 0; pat < 2; ++pat) {
            float best = 1e9;
            for (int rep = 0; rep < 6; ++rep) {
                (void)hipEventRecord(e0);
                if (pat == 0) k<0><<<2048, 256>>>(y, tpw, pitch, 0); else k<1><<<2048, 256>>>(y, tpw, pitch, 0);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0 && ms < best) best = ms;
            }
            const double bytes = (double)waves * tpw * 64 * 128;
            printf("pitch %d pattern %s: %.1f us, %.2f TB/s written (%.0f MB)\n", pitch, pat == 0 ? "16px x 64B" : "8px x 128B",
                   best * 1e3, bytes / best / 1e9, bytes / 1e6);
        }
        (void)hipFree(y);
    }
    return 0;
}

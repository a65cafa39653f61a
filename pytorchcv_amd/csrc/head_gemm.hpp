// Dense layer on a 1x1 map in fp32 (the classifier behind the global pool: nn.Linear of resnet.py:320-322, the 1x1 `output`
// convolution of mobilenetv2.py:196-199): y[n][j] = act(scale[j] * sum_k x[n][k] * W[j][k] + shift[j]).
// The implicit-GEMM kernel sees this layer as a 256 x 1000 x 2048 problem = 16 tiles of 128 x 128 on 256 CUs (140 us at batch 256);
// here a block of 4 waves owns 32 output channels x TN images, the waves split K in four contiguous ranges and meet in LDS in
// fixed order, so the grid is (Cout / 32) x (N / TN) blocks and an image's logits do not depend on its position in the batch or on
// the batch size. Operands go global -> registers (the 8 MB matrix streams from HBM once, the re-reads are L2 hits), 16 bytes per
// lane, in the k-association of `v_mfma_f32_16x16x4_f32`: lane (row l % 16, quarter l / 16) holds k = k0 + 4 * (l / 16) + t for the
// t-th of four MFMAs, for both operands. Exact fp32 products and sums (no reduced-precision step).
// Weights are read from the generic packed blob ([wrows][Kpad] fp32, rows in MFMA accumulator order - aux_kernels.hpp:58-61).
#pragma once
#include "pcv_common.hpp"

struct HeadParams {
    const float* x;        // [M][Xpitch]
    const float* w;        // packed rows [wrows][Kpad]
    const float* scale;    // [Cout] or NULL (= 1)
    const float* shift;    // [Cout] or NULL (= 0)
    float* y;              // [M][Ypitch]
    int M, K, Kpad, Cout, Xpitch, Ypitch, act;
};

template <int TN>
__global__ __launch_bounds__(256) void head_gemm_f32_kernel(const HeadParams p) {
    constexpr int NB = TN / 16;            // 16-image MFMA column blocks per wave
    constexpr int G = 4;                   // chunks (of 16 k) in flight per prefetch group
    __shared__ float part[4][32][TN + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int p0 = blockIdx.x * 32, n0 = blockIdx.y * TN;
    const int nchunk = p.K >> 4;
    const int c_lo = (int)((long)nchunk * wave / 4), c_hi = (int)((long)nchunk * (wave + 1) / 4);
    const float* arow[2];
    const float* brow[NB];
#pragma unroll
    for (int a = 0; a < 2; ++a) arow[a] = p.w + (size_t)(p0 + 16 * a + li) * p.Kpad + 4 * lq;
#pragma unroll
    for (int b = 0; b < NB; ++b) brow[b] = p.x + (size_t)min(n0 + 16 * b + li, p.M - 1) * p.Xpitch + 4 * lq;
    f32x4 acc[2][NB];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 av[2][G][2], bv[2][G][NB];       // [buffer][chunk of the group][fragment]
    auto load = [&](int buf, int c) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int cc = min(c + g, c_hi - 1);                  // past the range: a valid address, zeroed below
            const bool ok = c + g < c_hi;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(arow[a] + 16 * cc);
                av[buf][g][a] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int b = 0; b < NB; ++b) bv[buf][g][b] = *reinterpret_cast<const f32x4*>(brow[b] + 16 * cc);
        }
    };
    auto mac = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < NB; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[buf][g][a][t], bv[buf][g][b][t], acc[a][b], 0, 0, 0);
    };
    if (c_lo < c_hi) {
        load(0, c_lo);
        for (int c = c_lo; c < c_hi; c += 2 * G) {
            if (c + G < c_hi) load(1, c + G);
            mac(0);
            if (c + G < c_hi) {
                if (c + 2 * G < c_hi) load(0, c + 2 * G);
                mac(1);
            }
        }
    }
    // accumulator row r of lane l = channel position 4 * (l / 16) + r of the 16-row fragment = channel 8 * (l / 16) + 4 * a + r
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) part[wave][8 * lq + 4 * a + r][16 * b + li] = acc[a][b][r];
    __syncthreads();
    for (int o = threadIdx.x; o < 32 * TN; o += 256) {
        const int c = o & 31, n = o >> 5;
        const int ch = p0 + c;
        if (ch < p.Cout && n0 + n < p.M) {
            float v = ((part[0][c][n] + part[1][c][n]) + part[2][c][n]) + part[3][c][n];
            const float s = p.scale ? p.scale[ch] : 1.f, h = p.shift ? p.shift[ch] : 0.f;
            v = fmaf(v, s, h);
            p.y[(size_t)(n0 + n) * p.Ypitch + ch] = apply_act(v, p.act);
        }
    }
}

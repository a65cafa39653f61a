// dwconv.hpp - depthwise KSxKS convolution, NHWC, fused scale/shift/activation/residual epilogue.
//
// Replaces: dwconv_block -> ConvBlock(groups=out_channels) (reference pytorchcv/models/common/conv.py:437-473),
//           i.e. Conv2d(groups=C) + BatchNorm2d(eval) + ReLU6 of LinearBottleneck.conv2 (mobilenetv2.py:53-57).
//
// HBM-bound (3.4 FLOP/B): the job is to read every input byte once and write every output byte once with 16-byte
// accesses. One thread owns 8 consecutive channels (one 16-byte NHWC chunk for the 16-bit types) of one output
// column and walks DOWN the image keeping a KS x KS window of fp32 rows in registers, so per output row it loads
// only the STRIDE new input rows (KS chunks each). Consecutive lanes = consecutive channel chunks, then
// consecutive output columns: every wave-level load/store is one contiguous NHWC span; the +-1 column re-reads hit
// the same lines in the vector L1.
#pragma once
#include "pcv_common.hpp"

struct DwParams {
    const void* x;
    const void* w;        // packed [KS*KS][C] in DT
    const void* res;
    void* y;
    const float* scale;
    const float* shift;
    int N, H, W, C, Ho, Wo;
    int pt, pl;
    int C8;               // C / 8
    int TH;               // output rows per thread
    int nseg;             // ceil(Ho / TH)
    int act, post_act;
    long total;           // N * nseg * Wo * C8
};

template <int DT> __device__ __forceinline__ void load8(const void* base, size_t eidx, float (&v)[8]) {
    if constexpr (DT == PCV_F32) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + eidx);
        const f32x4 b = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + eidx + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
    } else {
        const u32x4 r = *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint16_t*>(base) + eidx);
#pragma unroll
        for (int e = 0; e < 4; ++e) unpack2<DT>(r[e], v[2 * e], v[2 * e + 1]);
    }
}
template <int DT> __device__ __forceinline__ void store8(void* base, size_t eidx, const float (&v)[8]) {
    if constexpr (DT == PCV_F32) {
        float* p = reinterpret_cast<float*>(base) + eidx;
        *reinterpret_cast<f32x4*>(p) = (f32x4){v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(p + 4) = (f32x4){v[4], v[5], v[6], v[7]};
    } else {
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pack2<DT>(v[2 * e], v[2 * e + 1]);
        *reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(base) + eidx) = o;
    }
}

template <int DT, int KS, int S>
__global__ __launch_bounds__(256) void dwconv_kernel(const DwParams p) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= p.total) return;
    const int c8 = (int)(idx % p.C8);
    long t = idx / p.C8;
    const int wo = (int)(t % p.Wo);
    t /= p.Wo;
    const int seg = (int)(t % p.nseg);
    const int n = (int)(t / p.nseg);
    const int c0 = c8 * 8;
    const int ho_begin = seg * p.TH;
    const int ho_end = min(p.Ho, ho_begin + p.TH);

    float wgt[KS * KS][8];
#pragma unroll
    for (int k = 0; k < KS * KS; ++k) load8<DT>(p.w, (size_t)k * p.C + c0, wgt[k]);
    float sc[8], sf[8];
    {
        const f32x4 a = *reinterpret_cast<const f32x4*>(p.scale + c0), b = *reinterpret_cast<const f32x4*>(p.scale + c0 + 4);
        const f32x4 c = *reinterpret_cast<const f32x4*>(p.shift + c0), d = *reinterpret_cast<const f32x4*>(p.shift + c0 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { sc[e] = a[e]; sc[4 + e] = b[e]; sf[e] = c[e]; sf[4 + e] = d[e]; }
    }

    const int wi0 = wo * S - p.pl;
    bool colok[KS];
#pragma unroll
    for (int q = 0; q < KS; ++q) colok[q] = (unsigned)(wi0 + q) < (unsigned)p.W;

    float win[KS][KS][8];
    auto load_row = [&](int hi, float (&row)[KS][8]) {
        const bool rowok = (unsigned)hi < (unsigned)p.H;
#pragma unroll
        for (int q = 0; q < KS; ++q) {
            if (rowok && colok[q]) {
                load8<DT>(p.x, (((size_t)n * p.H + hi) * p.W + (wi0 + q)) * p.C + c0, row[q]);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) row[q][e] = 0.f;
            }
        }
    };

    // prologue: the KS-S rows shared with the first output row
    int hi = ho_begin * S - p.pt;
#pragma unroll
    for (int r = 0; r < KS - S; ++r) load_row(hi + r, win[r + S]);

    for (int ho = ho_begin; ho < ho_end; ++ho, hi += S) {
        // slide: rows S..KS-1 become 0..KS-S-1, then load the S new rows at the bottom
#pragma unroll
        for (int r = 0; r < KS - S; ++r)
#pragma unroll
            for (int q = 0; q < KS; ++q)
#pragma unroll
                for (int e = 0; e < 8; ++e) win[r][q][e] = win[r + S][q][e];
#pragma unroll
        for (int r = KS - S; r < KS; ++r) load_row(hi + r, win[r]);

        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
        for (int r = 0; r < KS; ++r)
#pragma unroll
            for (int q = 0; q < KS; ++q)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = fmaf(win[r][q][e], wgt[r * KS + q][e], acc[e]);

        const size_t eoff = (((size_t)n * p.Ho + ho) * p.Wo + wo) * p.C + c0;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = apply_act(acc[e] * sc[e] + sf[e], p.act);
        if (p.res != nullptr) {
            float r8[8];
            load8<DT>(p.res, eoff, r8);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += r8[e];
        }
        if (p.post_act != PCV_ACT_NONE) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = apply_act(v[e], p.post_act);
        }
        store8<DT>(p.y, eoff, v);
    }
}

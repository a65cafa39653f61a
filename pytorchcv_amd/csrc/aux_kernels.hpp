// aux_kernels.hpp - the small HBM-bound kernels around the convolutions: layout conversion, weight packing,
// BN folding, max/avg pooling, SE squeeze / excite / scale. All NHWC, 8 channels (16 bytes at 16-bit) per thread.
#pragma once
#include "pcv_common.hpp"
#include "dwconv.hpp"   // load8 / store8

// ---- NCHW fp32 -> NHWC(cpitch, wpitch) --------------------------------------------------------------------
// What `net(x)` receives in the reference (resnet.py:333) is NCHW fp32; the hot path is NHWC. One thread = one
// output pixel x up to 8 channels; reads are coalesced along W per channel plane, pad channels/columns get zeros.
template <int OT>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x, void* __restrict__ y,
                                                          int N, int C, int H, int W, int cpitch, int wpitch,
                                                          uint32_t* __restrict__ ovf) {
    const long pix = (long)blockIdx.x * 256 + threadIdx.x;       // over N*H*wpitch
    const long npix = (long)N * H * wpitch;
    if (pix >= npix) return;
    const int c0 = blockIdx.y * 8;
    const int w = (int)(pix % wpitch);
    const long nh = pix / wpitch;
    const int h = (int)(nh % H);
    const int n = (int)(nh / H);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = c0 + e;
        v[e] = (c < C && w < W) ? x[(((size_t)n * C + c) * H + h) * W + w] : 0.f;
    }
    const size_t eoff = (size_t)pix * cpitch + c0;
    F16Guard<OT> guard;                                          // an image value beyond fp16's range
    guard.see(v);
    guard.commit(ovf);
    if (cpitch - c0 >= 8) {
        store8<OT>(y, eoff, v);
    } else if (cpitch - c0 == 4) {
        if constexpr (OT == PCV_F32) {
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(y) + eoff) = (f32x4){v[0], v[1], v[2], v[3]};
        } else {
            u32x2 o = {pack2<OT>(v[0], v[1]), pack2<OT>(v[2], v[3])};
            *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(y) + eoff) = o;
        }
    } else {
        for (int e = 0; e < cpitch - c0; ++e) store_elem<OT>(y, eoff + e, v[e]);
    }
}

// NHWC [N,H,W,C] -> NCHW fp32 (block-level API only)
template <int DT>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const void* __restrict__ x, float* __restrict__ y,
                                                          int N, int C, int H, int W, int cpitch) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;         // over N*C*H*W, w fastest
    const long total = (long)N * C * H * W;
    if (i >= total) return;
    const int w = (int)(i % W);
    long t = i / W;
    const int h = (int)(t % H);
    t /= H;
    const int c = (int)(t % C);
    const int n = (int)(t / C);
    y[i] = load_elem<DT>(x, (((size_t)n * H + h) * W + w) * cpitch + c);
}

// ---- weight packing (load time) ----------------------------------------------------------------------------
// ksrc[k] = (c_in_block | (r*kw+q) << 16) or 0xFFFFFFFF. Row position p of the packed matrix holds the output
// channel 32*(p>>5) + 8*((p&15)>>2) + 4*((p>>4)&1) + (p&3) of its group-block (MFMA accumulator-row order, so that
// the conv epilogue writes 8 consecutive channels per lane).
struct PackParams {
    const float* w;          // OIHW fp32
    void* out;               // [ngb][wrows][Kpad]
    const uint32_t* ksrc;    // [Kpad]
    int ngb, wrows, Kpad;
    int cout_blk, cin_blk;   // channels per group-block
    int Cg_in, Cg_out;       // channels per convolution group
    int khkw;
};
template <int DT>
__global__ __launch_bounds__(256) void pack_conv_kernel(const PackParams p) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)p.ngb * p.wrows * p.Kpad;
    if (i >= total) return;
    const int k = (int)(i % p.Kpad);
    long t = i / p.Kpad;
    const int pos = (int)(t % p.wrows);
    const int gb = (int)(t / p.wrows);
    const int ch = 32 * (pos >> 5) + 8 * ((pos & 15) >> 2) + 4 * ((pos >> 4) & 1) + (pos & 3);
    float v = 0.f;
    const uint32_t ks = p.ksrc[k];
    if (ch < p.cout_blk && ks != 0xFFFFFFFFu) {
        const int och = gb * p.cout_blk + ch;
        const int cin = gb * p.cin_blk + (int)(ks & 0xFFFFu);
        const int g = och / p.Cg_out;
        const int cl = cin - g * p.Cg_in;
        if (cl >= 0 && cl < p.Cg_in) v = p.w[((size_t)och * p.Cg_in + cl) * p.khkw + (ks >> 16)];
    }
    store_elem<DT>(p.out, (size_t)i, v);
}

// stem (stem_conv.hpp): w fp32 [Cout, Cin<=4, kh, kw] -> [kh][64 rows in MFMA order][8 pixels x 4 channels];
// window pixel `pix` of an output column is filter tap q = pix - qshift (qshift = pad_l & 1), anything else is zero.
template <int DT>
__global__ __launch_bounds__(256) void pack_stem_kernel(const float* __restrict__ w, void* __restrict__ out, int Cout, int Cin,
                                                       int kh, int kw, int qshift) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= kh * 64 * 32) return;
    const int k = i & 31, pos = (i >> 5) & 63, r = i >> 11;
    const int ch = 32 * (pos >> 5) + 8 * ((pos & 15) >> 2) + 4 * ((pos >> 4) & 1) + (pos & 3);
    const int pix = k >> 2, c = k & 3, q = pix - qshift;
    float v = 0.f;
    if (ch < Cout && c < Cin && q >= 0 && q < kw) v = w[(((size_t)ch * Cin + c) * kh + r) * kw + q];
    store_elem<DT>(out, (size_t)i, v);
}

// depthwise: w fp32 [C,1,kh,kw] -> [kh*kw][C]
template <int DT>
__global__ __launch_bounds__(256) void pack_dw_kernel(const float* __restrict__ w, void* __restrict__ out, int C, int khkw) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= C * khkw) return;
    const int c = i % C, k = i / C;
    store_elem<DT>(out, (size_t)i, w[(size_t)c * khkw + k]);
}

// dense 3x3 with 256 input channels, 16-bit (d3i_conv.hpp): the packed rows [row][Kpad] -> MFMA A fragments in the order a wave loads them,
// [64-row group][K-half kh = 0 .. Kpad / 32 - 1][16-row block nb][lane] 16 B: lane (fr, fq) holds row 64 grp + 16 nb + fr, K elements
// 32 kh + 8 fq .. + 7 - one coalesced 1 KB load per fragment.
__global__ __launch_bounds__(256) void pack_d3i_kernel(const u32x4* __restrict__ blob, u32x4* __restrict__ table, int Kpad, int total) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int lane = idx & 63, nb = (idx >> 6) & 3, nkh = Kpad / 32;
    const int kh = (idx >> 8) % nkh, grp = (idx >> 8) / nkh;
    const int row = 64 * grp + 16 * nb + (lane & 15);
    table[idx] = blob[((size_t)row * Kpad + 32 * kh + 8 * (lane >> 4)) / 8];
}

// depthwise 3x3, 16-bit: the packed taps [9][C] -> the compressed diagonal A fragments of the SPARSE matrix instruction
// (v_smfmac_f32_16x16x64, csrc/mbr.hpp MmaSp: the operand layout is documented and probed there), [chunk of 32 channels][half g]
// [filter row dy][lane] 16 B. Row i of a fragment is channel 32 c + 8 (i / 4) + 4 g + i % 4; B's lane quarter i / 4 supplies it at
// position i % 4 of its groups gb = 0 1 2 (left, centre, right tap; group 3 is padding): slot pair m of lane (i, q) is non-zero when
// it faces that quarter and a tap group. Built once at pack time; the fused inverted-residual kernel reads it as is.
__global__ __launch_bounds__(256) void pack_dw_sparse_kernel(const uint16_t* __restrict__ wd, u32x4* __restrict__ table, int C, int nChunks) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= nChunks * 6 * 64) return;
    const int l = idx & 63, dy = (idx >> 6) % 3, g = ((idx >> 6) / 3) & 1, c = (idx >> 6) / 6;
    const int i = l & 15, q = l >> 4;
    const int ch = 32 * c + 8 * (i >> 2) + 4 * g + (i & 3);
    u32x4 a4;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int kq = 2 * (q & 1) + (m >> 1), gb = 2 * (q >> 1) + (m & 1);
        const uint32_t w = (kq == (i >> 2) && gb < 3 && ch < C) ? wd[(3 * dy + gb) * C + ch] : 0u;
        a4[m] = (i & 3) == 3 ? w << 16 : w;                                         // kept positions (i % 4, 3), or (0, 3) for i % 4 == 3
    }
    table[idx] = a4;
}

// eval-mode BatchNorm2d (common/norm.py:34-50) folded to scale/shift, conv bias folded in
__global__ __launch_bounds__(256) void bn_fold_kernel(int C, const float* gamma, const float* beta, const float* mean,
                                                     const float* var, float eps, const float* bias, float* scale,
                                                     float* shift) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float s = 1.f, h = 0.f;
    if (gamma != nullptr) {
        s = gamma[c] / sqrtf(var[c] + eps);
        h = beta[c] - mean[c] * s;
    }
    if (bias != nullptr) h += bias[c] * s;
    scale[c] = s;
    shift[c] = h;
}

// ---- pooling ----------------------------------------------------------------------------------------------
// nn.MaxPool2d(k, s, p) (resnet.py:255-258): -inf padding, floor output size.
template <int DT>
__global__ __launch_bounds__(256) void maxpool_kernel(const void* __restrict__ x, void* __restrict__ y, int N, int H, int W,
                                                     int C, int Ho, int Wo, int k, int s, int pad) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int C8 = C / 8;
    const long total = (long)N * Ho * Wo * C8;
    if (i >= total) return;
    const int c0 = (int)(i % C8) * 8;
    long t = i / C8;
    const int wo = (int)(t % Wo);
    t /= Wo;
    const int ho = (int)(t % Ho);
    const int n = (int)(t / Ho);
    float m[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
    for (int r = 0; r < k; ++r) {
        const int hi = ho * s - pad + r;
        if ((unsigned)hi >= (unsigned)H) continue;
        for (int q = 0; q < k; ++q) {
            const int wi = wo * s - pad + q;
            if ((unsigned)wi >= (unsigned)W) continue;
            float v[8];
            load8<DT>(x, (((size_t)n * H + hi) * W + wi) * C + c0, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], v[e]);
        }
    }
    store8<DT>(y, (size_t)i * 8, m);
}

// nn.AvgPool2d(k, stride=s) without padding (resnet.py:316-318); general (non-global) case
template <int DT, int OT>
__global__ __launch_bounds__(256) void avgpool_kernel(const void* __restrict__ x, void* __restrict__ y, int N, int H, int W,
                                                     int C, int Ho, int Wo, int k, int s) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int C8 = C / 8;
    const long total = (long)N * Ho * Wo * C8;
    if (i >= total) return;
    const int c0 = (int)(i % C8) * 8;
    long t = i / C8;
    const int wo = (int)(t % Wo);
    t /= Wo;
    const int ho = (int)(t % Ho);
    const int n = (int)(t / Ho);
    float a[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = 0.f;
    for (int r = 0; r < k; ++r)
        for (int q = 0; q < k; ++q) {
            float v[8];
            load8<DT>(x, (((size_t)n * H + ho * s + r) * W + wo * s + q) * C + c0, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] += v[e];
        }
    const float inv = 1.f / (float)(k * k);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] *= inv;
    store8<OT>(y, (size_t)i * 8, a);
}

// Spatial mean over HW positions: global average pool (AvgPool2d(7) on a 7x7 map, resnet.py:316-318; AdaptiveAvgPool2d(1),
// efficientnet.py:339) and the SE squeeze (att.py:95). One 512-thread block per (image, group of <= 512 channel chunks):
// the threads tile [rows x chunks] so that a block reads one contiguous span of the NHWC map per iteration whatever the
// channel count (C = 32 keeps all 512 threads busy, 128 rows at a time), fp32 accumulation, rows meet in LDS.
template <int DT, int OT>
__global__ __launch_bounds__(512) void spatial_mean_kernel(const void* __restrict__ x, void* __restrict__ y, int HW, int C) {
    __shared__ float part[512][9];                       // +1: the row-sum reads below walk it with a stride of Gc rows
    const int n = blockIdx.x;
    const int C8 = C >> 3;
    const int g0 = blockIdx.y * 512;
    const int Gc = min(512, C8 - g0);                    // chunks of this group
    const int R = 512 / Gc;                              // rows in flight
    const int t = threadIdx.x;
    const int r = t / Gc, c = t - r * Gc;
    float a[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = 0.f;
    if (r < R) {
        const size_t base = (size_t)n * HW * C + (size_t)(g0 + c) * 8;
#pragma unroll 4
        for (int hw = r; hw < HW; hw += R) {
            float v[8];
            load8<DT>(x, base + (size_t)hw * C, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] += v[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) part[t][e] = a[e];
    __syncthreads();
    if (t < Gc) {
        const float inv = 1.f / (float)HW;
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = 0.f;
        for (int q = 0; q < R; ++q)
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] += part[q * Gc + t][e];
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] *= inv;
        store8<OT>(y, (size_t)n * C + (size_t)(g0 + t) * 8, a);
    }
}

// ---- SE excitation (att.py:96-103): gate = out_act(W2 . mid_act(W1 . mean + b1) + b2) -----------------------------------
// Both layers are the same small fp32 GEMM over the batch, out[n][j] = act(b[j] + sum_k W[j][k] * in[n][k]), launched
// twice. A block owns 8 images x TJ output rows, so every weight element fetched is used 8 times and the grid has
// (N / 8) x (J / TJ) blocks (one block per image re-read both matrices for every image and was the slowest kernel of
// MobileNetV3). Thread (jx, kp): output row jx of the block, K partition kp of 256 / TJ; inputs are LDS broadcasts
// staged 1024 columns at a time; the partitions meet in LDS.
__global__ __launch_bounds__(256) void se_fc_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                   const float* __restrict__ bias, float* __restrict__ out, int N, int K,
                                                   int J, int TJ, int act) {
    constexpr int IMG = 8, KC = 1024;
    __shared__ __attribute__((aligned(16))) float sin[IMG][KC];
    __shared__ float spart[256 * IMG];                           // [P][IMG][TJ]
    const int n0 = blockIdx.x * IMG;
    const int j0 = blockIdx.y * TJ;
    const int t = threadIdx.x;
    const int P = 256 / TJ;
    const int jx = t % TJ, kp = t / TJ;
    const int j = j0 + jx;
    const bool vec = (K & 3) == 0;
    float acc[IMG];
#pragma unroll
    for (int i = 0; i < IMG; ++i) acc[i] = 0.f;
    for (int kc = 0; kc < K; kc += KC) {
        const int kn = min(KC, K - kc);
        if (kc > 0) __syncthreads();
        for (int i = t; i < IMG * kn; i += 256) {
            const int img = i / kn, k = i - img * kn;
            sin[img][k] = n0 + img < N ? in[(size_t)(n0 + img) * K + kc + k] : 0.f;
        }
        __syncthreads();
        if (j < J) {
            const float* wr = w + (size_t)j * K + kc;
            if (vec) {
                const int kper = ((kn / 4 + P - 1) / P) * 4;
                const int kb = min(kn, kp * kper), ke = min(kn, kb + kper);
                for (int k = kb; k < ke; k += 4) {
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + k);
#pragma unroll
                    for (int i = 0; i < IMG; ++i) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(&sin[i][k]);
                        // explicit fma chain: every image slot must round identically (the compiler packs slot pairs into
                        // v_pk_* ops; with `a*b + c*d + ...` some pairs got fused and others mul+add: 1-ulp batch-position dependence)
                        acc[i] = fmaf(wv[3], v[3], fmaf(wv[2], v[2], fmaf(wv[1], v[1], fmaf(wv[0], v[0], acc[i]))));
                    }
                }
            } else {
                const int kper = (kn + P - 1) / P;
                const int kb = min(kn, kp * kper), ke = min(kn, kb + kper);
                for (int k = kb; k < ke; ++k) {
                    const float wv = wr[k];
#pragma unroll
                    for (int i = 0; i < IMG; ++i) acc[i] = fmaf(wv, sin[i][k], acc[i]);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < IMG; ++i) spart[(kp * IMG + i) * TJ + jx] = acc[i];
    __syncthreads();
    for (int o = t; o < IMG * TJ; o += 256) {
        const int i = o / TJ, r = o - i * TJ;
        if (j0 + r < J && n0 + i < N) {
            float a = bias[j0 + r];
            for (int q = 0; q < P; ++q) a += spart[(q * IMG + i) * TJ + r];
            out[(size_t)(n0 + i) * J + j0 + r] = apply_act(a, act);
        }
    }
}

// y = post_act(x * gate[n,c] + residual)  (att.py:104, seresnet.py:69-71)
template <int DT>
__global__ __launch_bounds__(256) void se_scale_kernel(const void* __restrict__ x, const float* __restrict__ gate,
                                                      const void* __restrict__ res, void* __restrict__ y, long total8,
                                                      int HW, int C, int post_act, uint32_t* __restrict__ ovf) {
    const int C8 = C / 8;
    const ActClamp pact = make_act(post_act);
    F16Guard<DT> guard;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total8; i += (long)gridDim.x * 256) {
        const int c0 = (int)(i % C8) * 8;
        const long pix = i / C8;
        const long n = pix / HW;
        float v[8], g[8];
        load8<DT>(x, (size_t)i * 8, v);
        load8<PCV_F32>(gate, (size_t)n * C + c0, g);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= g[e];
        if (res != nullptr) {
            float r[8];
            load8<DT>(res, (size_t)i * 8, r);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += r[e];
        }
        if (post_act != PCV_ACT_NONE) apply_act8(v, pact);
        guard.see(v);
        store8<DT>(y, (size_t)i * 8, v);
    }
    guard.commit(ovf);
}

// y = act(x * scale[c] + shift[c]): the BatchNorm + activation in FRONT of a convolution (PreConvBlock, conv.py:776-779;
// PreResActivation, preresnet.py:199-222) when no producing convolution can take it as its epilogue. HBM-bound, 16 B/lane.
template <int DT>
__global__ __launch_bounds__(256) void bn_act_kernel(const void* __restrict__ x, const float* __restrict__ scale,
                                                    const float* __restrict__ shift, void* __restrict__ y, long total8,
                                                    int C, int xpitch, int act_code, uint32_t* __restrict__ ovf) {
    const int C8 = C / 8;
    const ActClamp act = make_act(act_code);
    F16Guard<DT> guard;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total8; i += (long)gridDim.x * 256) {
        const int c0 = (int)(i % C8) * 8;
        float v[8], a[8], b[8];
        load8<DT>(x, (size_t)(i / C8) * xpitch + c0, v);          // x may be the leading channels of a wider (concat) tensor
        load8<PCV_F32>(scale, c0, a);
        load8<PCV_F32>(shift, c0, b);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] * a[e] + b[e];
        apply_act8(v, act);
        guard.see(v);
        store8<DT>(y, (size_t)i * 8, v);
    }
    guard.commit(ovf);
}

// ---- ImageNet preprocessing: uint8 HWC image batch -> normalised network input in one pass --------------------------------
// The reference's models expect "ordinary normalization" of a centre crop (README.md:12-13; model_metainfos.csv columns
// img_size = 224, img_scale = 0.875): x = (u8 / 255 - mean[c]) / std[c]. What the stem kernel wants is NHWC with the channels
// padded to 4 and the row pitch even, in the compute type; producing that directly from the decoded uint8 frames folds
// crop + normalise + layout + cast into one read of 1 byte per element (instead of an fp32 NCHW tensor written by the host
// pipeline and re-read by nchw_to_nhwc). One thread = one output pixel.
template <int OT>
__global__ __launch_bounds__(256) void preprocess_u8_kernel(const unsigned char* __restrict__ x, void* __restrict__ y, int N,
                                                           int Hs, int Ws, int C, int top, int left, int H, int W, int cpitch,
                                                           int wpitch, const float* __restrict__ mean,
                                                           const float* __restrict__ inv_std, uint32_t* __restrict__ ovf) {
    const long pix = (long)blockIdx.x * 256 + threadIdx.x;       // over N*H*wpitch
    if (pix >= (long)N * H * wpitch) return;
    const int w = (int)(pix % wpitch);
    const long nh = pix / wpitch;
    const int h = (int)(nh % H);
    const int n = (int)(nh / H);
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (w < W) {
        const unsigned char* src = x + (((size_t)n * Hs + top + h) * Ws + left + w) * C;
        for (int c = 0; c < C && c < 4; ++c) v[c] = ((float)src[c] * (1.f / 255.f) - mean[c]) * inv_std[c];
    }
    const size_t eoff = (size_t)pix * cpitch;
    F16Guard<OT> guard;
    guard.see(v);
    guard.commit(ovf);
    if constexpr (OT == PCV_F32) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(y) + eoff) = (f32x4){v[0], v[1], v[2], v[3]};
    } else {
        u32x2 o = {pack2<OT>(v[0], v[1]), pack2<OT>(v[2], v[3])};
        *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(y) + eoff) = o;
    }
}

// ---- channel plumbing of ShuffleNet-style units (shufflenetv2.py:69-91; common/tutti.py:267-291) ---------------------------------
// torch.chunk(x, 2, dim=1)[1]: y[.., i] = x[.., off + i], i < C; y has `ypitch` physical channels, the pads are written as zero.
// Channel offsets here are arbitrary (58 of 116), so this works element-wise: one thread = one pixel x 8 output channels.
template <int DT>
__global__ __launch_bounds__(256) void channel_slice_kernel(const void* __restrict__ x, void* __restrict__ y, long rows, int C,
                                                           int off, int xpitch, int ypitch) {
    const int P8 = ypitch / 8;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * P8) return;
    const long row = i / P8;
    const int c0 = (int)(i - row * P8) * 8;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = c0 + e < C ? load_elem<DT>(x, (size_t)row * xpitch + off + c0 + e) : 0.f;
    store8<DT>(y, (size_t)row * ypitch + c0, v);
}
// torch.cat along channels, the copy form (arch.py:93-94 when a branch cannot write its slice itself): y[row, off + c] = x[row, c],
// c < C. C, off and both pitches are multiples of 8: one thread = one 16-byte chunk.
template <int DT>
__global__ __launch_bounds__(256) void channel_concat_kernel(const void* __restrict__ x, void* __restrict__ y, long rows, int C,
                                                            int xpitch, int ypitch, int off) {
    const int C8 = C / 8;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * C8) return;
    const long row = i / C8;
    const int c0 = (int)(i - row * C8) * 8;
    constexpr int ES = Elem<DT>::BYTES;
    const char* src = reinterpret_cast<const char*>(x) + ((size_t)row * xpitch + c0) * ES;
    char* dst = reinterpret_cast<char*>(y) + ((size_t)row * ypitch + off + c0) * ES;
    *reinterpret_cast<u32x4*>(dst) = *reinterpret_cast<const u32x4*>(src);
    if constexpr (ES == 4) *reinterpret_cast<u32x4*>(dst + 16) = *reinterpret_cast<const u32x4*>(src + 16);
}

// F.interpolate(mode = "bilinear" | "nearest", align_corners) of InterpolationBlock (tutti.py:232-246) on NHWC: one thread = one
// output pixel x 8 channels, fp32 arithmetic. Source coordinates as ATen computes them: align_corners: dst * (in - 1) / (out - 1);
// otherwise max(0, (dst + 0.5) * in / out - 0.5); nearest: min(floor(dst * in / out), in - 1).
template <int DT>
__global__ __launch_bounds__(256) void interpolate_kernel(const void* __restrict__ x, void* __restrict__ y, int N, int H, int W,
                                                         int C, int Ho, int Wo, int bilinear, int align_corners) {
    const int C8 = C / 8;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)N * Ho * Wo * C8) return;
    const int c0 = (int)(i % C8) * 8;
    long t = i / C8;
    const int wo = (int)(t % Wo);
    t /= Wo;
    const int ho = (int)(t % Ho);
    const int n = (int)(t / Ho);
    float v[8];
    if (!bilinear) {
        const float sh = (float)H / (float)Ho, sw = (float)W / (float)Wo;
        const int hi = min((int)floorf(ho * sh), H - 1), wi = min((int)floorf(wo * sw), W - 1);
        load8<DT>(x, (((size_t)n * H + hi) * W + wi) * C + c0, v);
    } else {
        float fh, fw;
        if (align_corners) {
            fh = Ho > 1 ? ho * ((float)(H - 1) / (float)(Ho - 1)) : 0.f;
            fw = Wo > 1 ? wo * ((float)(W - 1) / (float)(Wo - 1)) : 0.f;
        } else {
            fh = fmaxf(((float)ho + 0.5f) * ((float)H / (float)Ho) - 0.5f, 0.f);
            fw = fmaxf(((float)wo + 0.5f) * ((float)W / (float)Wo) - 0.5f, 0.f);
        }
        const int h0 = min((int)fh, H - 1), w0 = min((int)fw, W - 1);
        const int h1 = min(h0 + 1, H - 1), w1 = min(w0 + 1, W - 1);
        const float lh = fh - (float)h0, lw = fw - (float)w0;
        float a[8], b[8], c[8], d[8];
        load8<DT>(x, (((size_t)n * H + h0) * W + w0) * C + c0, a);
        load8<DT>(x, (((size_t)n * H + h0) * W + w1) * C + c0, b);
        load8<DT>(x, (((size_t)n * H + h1) * W + w0) * C + c0, c);
        load8<DT>(x, (((size_t)n * H + h1) * W + w1) * C + c0, d);
#pragma unroll
        for (int e = 0; e < 8; ++e)
            v[e] = (1.f - lh) * ((1.f - lw) * a[e] + lw * b[e]) + lh * ((1.f - lw) * c[e] + lw * d[e]);
    }
    store8<DT>(y, (size_t)i * 8, v);
}

// torch.cat((a, b), dim=1) followed by channel_shuffle(groups = 2): y[.., 2i] = a[.., i], y[.., 2i + 1] = b[.., i], i < Ch.
template <int DT>
__global__ __launch_bounds__(256) void channel_interleave2_kernel(const void* __restrict__ a, const void* __restrict__ b,
                                                                 void* __restrict__ y, long rows, int Ch, int apitch, int bpitch,
                                                                 int ypitch) {
    const int P8 = ypitch / 8;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * P8) return;
    const long row = i / P8;
    const int c0 = (int)(i - row * P8) * 8;          // even
    float v[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int j = c0 / 2 + e;
        v[2 * e] = j < Ch ? load_elem<DT>(a, (size_t)row * apitch + j) : 0.f;
        v[2 * e + 1] = j < Ch ? load_elem<DT>(b, (size_t)row * bpitch + j) : 0.f;
    }
    store8<DT>(y, (size_t)row * ypitch + c0, v);
}


// ---- fp16 range guard: begin / end of a guarded forward (pcv_fp16_guard_begin / _end) -----------------------------------------------
// begin: remember the overflow counter; end: if it moved, the forward rounded something beyond fp16's range - its fp32 outputs
// become NaN (a loud failure instead of plausible numbers; the caller reruns in bf16).
__global__ void f16_guard_begin_kernel(const uint32_t* __restrict__ counter, uint32_t* __restrict__ slot) { *slot = *counter; }
__global__ __launch_bounds__(256) void f16_guard_end_kernel(const uint32_t* __restrict__ counter, const uint32_t* __restrict__ slot,
                                                           float* __restrict__ y, long count) {
    if (*counter == *slot) return;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long)gridDim.x * 256) y[i] = __builtin_nanf("");
}

// igemm_conv.hpp - im2col-free implicit-GEMM convolution on gfx950 MFMA with a fused BN/activation/residual
// epilogue. One kernel template serves dense 1x1 / 3x3 / kxk, strided, dilated, asymmetric-padded, stem
// (Cin <= 4, pixel-pair chunks) and block-diagonal grouped convolutions.
//
// Replaces: nn.Conv2d + nn.BatchNorm2d(eval) + activation of ConvBlock.forward
//           (reference pytorchcv/models/common/conv.py:278-286) and the unit's residual add + activation
//           (resnet.py:227-228, resnext.py:114-115, mobilenetv2.py:69-70).
//
// GEMM view (operands swapped so that channels land on accumulator rows and the epilogue writes 16-byte
// channel-contiguous NHWC pieces):
//     Y^T[ch, pixel] = sum_k Wp[ch, k] * X[pixel, k],  k = (filter row r, filter col q, input channel c)
//   * "A" operand = packed weights Wp[Cout_pad][Kpad] (K contiguous, rows in MFMA order, see pack kernel)
//   * "B" operand = NHWC activations gathered on the fly: row = output pixel m=(n,ho,wo), 16-byte chunk j of the
//     K axis = CE consecutive input channels of input pixel (ho*s-p+dy_j, wo*s-p+dx_j).
//
// Data movement: both tiles go global -> LDS with `buffer_load_dwordx4 ... lds` (LDS-DMA, no VGPR round trip).
// The per-lane SOURCE offset implements the im2col gather; a padded (out-of-image) tap is an offset beyond the
// buffer's num_records, for which the hardware writes zeros. LDS rows are 128 B (8 chunks); chunk slot s of row
// r holds K-chunk s ^ (r & 7) (swizzle applied on the source side, LDS image stays lane-linear) so that the
// `ds_read_b128` fragment reads of 16 consecutive rows are bank-conflict free.
//
// Pipeline: 2 LDS stages; per K-step one barrier: { wait DMA(t) ; barrier ; issue DMA(t+1) ; MFMA(t) }.
#pragma once
#include "pcv_common.hpp"

#define IGEMM_MAX_TAPS 16

struct IgemmParams {
    const void* x;          // NHWC activations
    const void* w;          // packed weights (weights region of the blob)
    const uint32_t* ktab;   // 2 dwords per K-chunk: {c0 | r<<16 | q<<20, (int16)dy | (int16)dx<<16}
    const void* res;        // residual NHWC [M, Cout_total] or null
    void* y;                // NHWC [M, Cout_total]
    const float* scale;     // [Cout_total]
    const float* shift;
    uint32_t x_bytes;       // buffer num_records for x
    uint32_t w_bytes;       // buffer num_records for w
    int M;                  // N*Ho*Wo
    int Cout;               // valid output channels per group-block (== Cout_total when gridDim.y == 1)
    int Cout_total;         // channel pitch of y / residual
    int cout_blk;           // output channels per blockIdx.y step
    int cin_blk;            // input-channel offset per blockIdx.y step
    int wrows_blk;          // packed weight rows per blockIdx.y step
    FastDiv div_howo, div_wo;
    int HoWo, Wo;
    int H, W, Wpitch, Cpitch;
    int sh, sw, pt, pl;
    int nR, nQ;
    int dy[IGEMM_MAX_TAPS];
    int dx[IGEMM_MAX_TAPS];
    int nk;                 // K steps of 128 B
    int Kpad;               // packed row length in elements
    int act, post_act;
    int nPixTiles, nChTiles;
};

template <int DT> struct Mma;
template <> struct Mma<PCV_BF16> {
    typedef s16x8 frag;
    static __device__ __forceinline__ f32x4 run(const frag& a, const frag& b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mma<PCV_F16> {
    typedef f16x8 frag;
    static __device__ __forceinline__ f32x4 run(const frag& a, const frag& b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mma<PCV_F32> {
    // exact-f32 MFMA (v_mfma_f32_16x16x4_f32): the 16-byte chunk held by lane group q carries k = 4q+e, and
    // MFMA e (0..3) sums element e over the four lane groups - the same k association on both operands.
    typedef f32x4 frag;
    static __device__ __forceinline__ f32x4 run(const frag& a, const frag& b, f32x4 c) {
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c, 0, 0, 0);
        return c;
    }
};

// DT: storage type of x / w / residual.  OT: storage type of y.  CB: 16-channel blocks per wave (2 or 4).
// PB: 16-pixel blocks per wave.  WC x WP: wave grid (channels x pixels).  RAGGED: Cout not a multiple of 8.
template <int DT, int OT, int CB, int PB, int WC, int WP, bool RAGGED>
__global__ __launch_bounds__(64 * WC * WP) void igemm_conv_kernel(const IgemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass only needs the launch stub (buffer-resource types are device-only)
    constexpr int NW = WC * WP;
    constexpr int BM = 16 * CB * WC;          // channel rows per block tile
    constexpr int BP = 16 * PB * WP;          // pixel rows per block tile
    constexpr int ES = Elem<DT>::BYTES;
    constexpr int CE = 16 / ES;               // elements per 16-byte chunk
    constexpr int STAGE = (BM + BP) * 128;    // bytes per LDS stage
    constexpr int WLOADS = BM / (8 * NW);     // LDS-DMA wave-instructions per thread for the weight tile
    constexpr int XLOADS = BP / (8 * NW);
    static_assert(BM % (8 * NW) == 0 && BP % (8 * NW) == 0, "tile rows must split evenly over the waves");
    typedef typename Mma<DT>::frag frag;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave / WP, wp = wave % WP;

    // ---- which tile -------------------------------------------------------------------------------------
    const uint32_t tile = xcd_remap(blockIdx.x, gridDim.x);
    const int chTile = tile % p.nChTiles;
    const int pixTile = tile / p.nChTiles;
    const int gb = blockIdx.y;
    const int tileP0 = pixTile * BP;

    // ---- per-thread gather state: XLOADS pixel rows, one K-chunk column --------------------------------
    const int lrow = lane >> 3;               // row within an 8-row DMA piece
    const int cs = (lane & 7) ^ lrow;         // K-chunk this lane fetches (source-side swizzle)
    int rbase[XLOADS];
    uint32_t rmask[XLOADS];
#pragma unroll
    for (int i = 0; i < XLOADS; ++i) {
        const int prow = 8 * (i * NW + wave) + lrow;
        const int m = tileP0 + prow;
        uint32_t mask = 0;
        int base = 0;
        if (m < p.M) {
            const uint32_t n = fastdiv((uint32_t)m, p.div_howo);
            const uint32_t rem = (uint32_t)m - n * (uint32_t)p.HoWo;
            const uint32_t ho = fastdiv(rem, p.div_wo);
            const uint32_t wo = rem - ho * (uint32_t)p.Wo;
            const int hi0 = (int)ho * p.sh - p.pt;
            const int wi0 = (int)wo * p.sw - p.pl;
            base = (((int)n * p.H + hi0) * p.Wpitch + wi0) * p.Cpitch + gb * p.cin_blk;
            for (int r = 0; r < p.nR; ++r)
                mask |= ((uint32_t)(hi0 + p.dy[r]) < (uint32_t)p.H ? 1u : 0u) << r;
            for (int q = 0; q < p.nQ; ++q)
                mask |= ((uint32_t)(wi0 + p.dx[q]) < (uint32_t)p.W ? 1u : 0u) << (16 + q);
        }
        rbase[i] = base;
        rmask[i] = mask;
    }
    // weight rows: loop-invariant byte offsets, the K-step advance goes through the scalar offset
    uint32_t woff[WLOADS];
#pragma unroll
    for (int i = 0; i < WLOADS; ++i) {
        const int wrow = 8 * (i * NW + wave) + lrow;
        woff[i] = (uint32_t)(((gb * p.wrows_blk + chTile * BM + wrow) * p.Kpad + cs * CE) * ES);
    }

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);

    auto load_kdesc = [&](int t) -> u32x2 {
        return *reinterpret_cast<const u32x2*>(p.ktab + 2 * (t * 8 + cs));
    };
    auto stage = [&](int t, int buf, u32x2 kd) {
        char* sbase = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < WLOADS; ++i) {
            char* dst = sbase + (8 * (i * NW + wave)) * 128;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, PCV_LDS(dst), 16, woff[i], t * 128, 0, 0);
        }
        const int c0 = (int)(kd[0] & 0xFFFFu);
        const uint32_t r = (kd[0] >> 16) & 15u, q = (kd[0] >> 20) & 15u;
        const int dy = (int)(short)(kd[1] & 0xFFFFu), dx = (int)(short)(kd[1] >> 16);
        const int koff = (dy * p.Wpitch + dx) * p.Cpitch + c0;
#pragma unroll
        for (int i = 0; i < XLOADS; ++i) {
            char* dst = sbase + (BM + 8 * (i * NW + wave)) * 128;
            const bool ok = ((rmask[i] >> r) & (rmask[i] >> (16 + q)) & 1u) != 0;
            const uint32_t voff = ok ? (uint32_t)((rbase[i] + koff) * ES) : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, PCV_LDS(dst), 16, voff, 0, 0, 0);
        }
    };

    // ---- fragment read addresses ------------------------------------------------------------------------
    const int fr = lane & 15, fq = lane >> 4;
    const int swz0 = ((fq) ^ (fr & 7)) << 4;
    const int swz1 = ((fq + 4) ^ (fr & 7)) << 4;
    const int wfrag = (wc * 16 * CB + fr) * 128;
    const int xfrag = (BM + wp * 16 * PB + fr) * 128;

    f32x4 acc[CB][PB];
#pragma unroll
    for (int i = 0; i < CB; ++i)
#pragma unroll
        for (int j = 0; j < PB; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- main loop ----------------------------------------------------------------------------------------
    u32x2 kd = load_kdesc(0);
    stage(0, 0, kd);
    if (p.nk > 1) kd = load_kdesc(1);
    for (int t = 0; t < p.nk; ++t) {
        __syncthreads();                       // DMA(t) landed for every wave; stage (t+1)&1 is free again
        if (t + 1 < p.nk) {
            stage(t + 1, (t + 1) & 1, kd);
            if (t + 2 < p.nk) kd = load_kdesc(t + 2);
        }
        const char* sbase = smem + (t & 1) * STAGE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int swz = kk == 0 ? swz0 : swz1;
            frag a[CB], b[PB];
#pragma unroll
            for (int i = 0; i < CB; ++i) a[i] = *reinterpret_cast<const frag*>(sbase + wfrag + i * 2048 + swz);
#pragma unroll
            for (int j = 0; j < PB; ++j) b[j] = *reinterpret_cast<const frag*>(sbase + xfrag + j * 2048 + swz);
#pragma unroll
            for (int i = 0; i < CB; ++i)
#pragma unroll
                for (int j = 0; j < PB; ++j) acc[i][j] = Mma<DT>::run(a[i], b[j], acc[i][j]);
        }
    }

    // ---- epilogue: scale/shift -> act -> (+residual) -> post_act -> NHWC store ----------------------------
    // Packed weight row (16*i + rho) of a 64-row group holds channel 32*(i>>1) + 8*(rho>>2) + 4*(i&1) + (rho&3),
    // so lane group fq owns the 8 consecutive channels 32*ip + 8*fq .. +7 (accumulators 2ip and 2ip+1).
    const int chBlk = chTile * BM + wc * 16 * CB;       // first channel (within the group-block) of this wave
    const int chGlob0 = gb * p.cout_blk;
#pragma unroll
    for (int ip = 0; ip < CB / 2; ++ip) {
        const int ch0 = chBlk + 32 * ip + 8 * fq;       // within the group-block
        if (ch0 >= p.Cout) continue;
        const int chg = chGlob0 + ch0;                  // global channel
        float sc[8], sf[8];
        if constexpr (RAGGED) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool ok = ch0 + e < p.Cout;
                sc[e] = ok ? (p.scale ? p.scale[chg + e] : 1.f) : 0.f;
                sf[e] = ok ? (p.shift ? p.shift[chg + e] : 0.f) : 0.f;
            }
        } else {
            f32x4 s0 = {1.f, 1.f, 1.f, 1.f}, s1 = s0, h0 = {0.f, 0.f, 0.f, 0.f}, h1 = h0;
            if (p.scale != nullptr) {
                s0 = *reinterpret_cast<const f32x4*>(p.scale + chg);
                s1 = *reinterpret_cast<const f32x4*>(p.scale + chg + 4);
            }
            if (p.shift != nullptr) {
                h0 = *reinterpret_cast<const f32x4*>(p.shift + chg);
                h1 = *reinterpret_cast<const f32x4*>(p.shift + chg + 4);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) { sc[e] = s0[e]; sc[4 + e] = s1[e]; sf[e] = h0[e]; sf[4 + e] = h1[e]; }
        }
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            const int m = tileP0 + wp * 16 * PB + 16 * j + fr;
            if (m >= p.M) continue;
            const size_t eoff = (size_t)m * p.Cout_total + chg;
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = acc[2 * ip][j][e] * sc[e] + sf[e];
                v[4 + e] = acc[2 * ip + 1][j][e] * sc[4 + e] + sf[4 + e];
            }
            if (p.act != PCV_ACT_NONE) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = apply_act(v[e], p.act);
            }
            if (p.res != nullptr) {
                if constexpr (RAGGED) {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (ch0 + e < p.Cout) v[e] += load_elem<DT>(p.res, eoff + e);
                } else if constexpr (DT == PCV_F32) {
                    const f32x4 r0 = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.res) + eoff);
                    const f32x4 r1 = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.res) + eoff + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] += r0[e]; v[4 + e] += r1[e]; }
                } else {
                    const u32x4 r = *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint16_t*>(p.res) + eoff);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float lo, hi;
                        unpack2<DT>(r[e], lo, hi);
                        v[2 * e] += lo;
                        v[2 * e + 1] += hi;
                    }
                }
            }
            if (p.post_act != PCV_ACT_NONE) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = apply_act(v[e], p.post_act);
            }
            if constexpr (RAGGED) {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (ch0 + e < p.Cout) store_elem<OT>(p.y, eoff + e, v[e]);
            } else if constexpr (OT == PCV_F32) {
                float* yp = reinterpret_cast<float*>(p.y) + eoff;
                *reinterpret_cast<f32x4*>(yp) = (f32x4){v[0], v[1], v[2], v[3]};
                *reinterpret_cast<f32x4*>(yp + 4) = (f32x4){v[4], v[5], v[6], v[7]};
            } else {
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = pack2<OT>(v[2 * e], v[2 * e + 1]);
                *reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(p.y) + eoff) = o;
            }
        }
    }
#endif  // __HIP_DEVICE_COMPILE__
}

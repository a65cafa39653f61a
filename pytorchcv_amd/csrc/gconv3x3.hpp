// gconv3x3.hpp - grouped 3x3 / stride 1 / pad 1 convolution with 4, 8 or 16 channels per group (ResNeXt 32x4d stages 1-3,
// SE-ResNeXt; reference resnext.py:62-80 via conv3x3_block(groups=cardinality), common/conv.py:340-386), 16-bit storage.
//
// On the generic implicit GEMM a grouped convolution is a block-diagonal 32x32 GEMM per tap: every pixel's 64-byte channel
// segment is gathered nine times from L2 and 50-88 % of the MFMA work multiplies zeros; the layers reach 1.5 TB/s (0.2 of the
// HBM peak) although they are pure streaming (3x3 x 4..16 MACs per output). Here:
//   * a block owns 64 channels (whole groups) x 128 pixels; the activation tile with its halo (flat pixel range
//     [p0 - WPAD, p0 + 128 + WPAD) of 128-byte rows: taps are row shifts (r-1) W + (q-1), borders are
//     resolved by zeroing the B fragment of a lane whose pixel leaves the image) is staged ONCE by LDS-DMA, double buffered;
//   * the K axis of one MFMA is (2 taps) x (16 input channels): a 16-channel slab (= 4, 2 or 1 whole groups) needs 5 K-steps for
//     its 9 taps (the 10th is zero weight) instead of 18 K-steps x 4 row fragments per 64 channels; inside a slab the weight
//     fragment is block-diagonal over its groups. Lane (pixel fr, K-chunk fq) reads tap 2 ks + (fq >> 1), channels
//     8 (fq & 1) .. +8 of the slab straight from the halo tile (per-lane LDS addresses);
//   * all weights of the 64-channel block (4 slabs x 5 K-steps) stay in registers while consecutive tiles keep the same
//     channel block (tile order: channel block slowest).
// HBM-bound by construction: one 128-byte read and one write per pixel and 64 channels (+ halo rows, L2 hits).
#pragma once
#include "pcv_common.hpp"
#include "igemm_conv.hpp"     // Mma<DT>, FastDiv

struct GConvParams {
    const void* x;
    const void* w;          // packed [C / 16 slabs][5 K-steps][16 rows][32 K] (K = tap-in-pair x 16 input channels of the slab)
    void* y;
    const float* scale;
    const float* shift;
    uint32_t x_bytes, w_bytes, y_bytes;
    int M, H, W, C, HW;
    FastDiv div_hw, div_w;
    int nPixTiles, nTiles;
    int act;
    uint32_t* ovf;          // the context's fp16 overflow counter (pcv_common.hpp, F16Guard)
};

template <int N> __device__ __forceinline__ void gconv_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int WPAD> struct GConvCfg {
    static constexpr int BP = 128, XR = BP + 2 * WPAD, XL = XR / 32;       // tile rows, LDS-DMA pieces per thread
    static constexpr int LDS = 2 * XR * 128;
};

template <int DT, int WPAD>
__global__ __launch_bounds__(256, 2) void gconv3x3_kernel(const GConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef GConvCfg<WPAD> G;
    constexpr int BP = G::BP, XR = G::XR, XL = G::XL, NW = 4, PBW = 2;
    static_assert(DT != PCV_F32 && XR % 32 == 0, "16-bit storage; the tile rows split evenly into 8-row pieces over 4 waves");
    typedef typename Mma<DT>::frag frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [X: 2 x XR rows of 128 B]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane >> 3;
    const int cs_lane = (lane & 7) ^ lrow;
    const int fr = lane & 15, fq = lane >> 4;
    const int thalf = fq >> 1, chalf = fq & 1;         // which tap of the K-step's pair / which 8 of the slab's 16 channels

    const int perXcd = (p.nTiles + 7) >> 3;
    const int xcd = blockIdx.x & 7;
    const int tstride = gridDim.x >> 3;
    int tile = xcd * perXcd + (int)(blockIdx.x >> 3);
    const int tend = min(p.nTiles, (xcd + 1) * perXcd);
    if (tile >= tend) return;

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);

    // tile t = (channel block t / nPixTiles, pixel tile t % nPixTiles): a block mostly stays inside one channel block
    auto issue_x = [&](int t, int xb, bool live) {
        char* xdst = smem + xb * (XR * 128);
        const int cb = t / p.nPixTiles;
        const int p0 = (t - cb * p.nPixTiles) * BP;
#pragma unroll
        for (int j = 0; j < XL; ++j) {
            const int c = p0 - WPAD + 8 * (j * NW + wave) + lrow;
            const uint32_t off = (live && c >= 0 && c < p.M) ? (uint32_t)((c * p.C + cb * 64 + cs_lane * 8) * 2) : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, PCV_LDS(xdst + (8 * (j * NW + wave)) * 128), 16, off, 0, 0, 0);
        }
    };
    const int xrow0 = WPAD + wave * (16 * PBW) + fr;   // tile row of this lane's pixel (block j adds 16 j)
    const ActClamp act = make_act(p.act);

    frag a[4][5];                                      // [slab of the channel block][K-step]
    f32x4 sc[4], sf[4];                                // channels 64 cb + 16 i + 4 fq .. +4 (MFMA rows 4 fq + e)
    int cur_cb = -1;

    issue_x(tile, 0, true);
    int xb = 0;
    while (true) {
        const int ntile = tile + tstride;
        const bool has_next = ntile < tend;
        const int cb = tile / p.nPixTiles;
        const int p0 = (tile - cb * p.nPixTiles) * BP;
        bool reloaded = false;
        if (cb != cur_cb) {                            // uniform: new channel block -> its weights and BN constants
            cur_cb = cb;
            reloaded = true;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int ks = 0; ks < 5; ++ks)
                    a[i][ks] = __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(
                        wrsrc, (uint32_t)(((((cb * 4 + i) * 5 + ks) * 16 + fr) * 32 + 8 * fq) * 2), 0, 0));
                const int ch = cb * 64 + 16 * i + 4 * fq;
                sc[i] = p.scale ? *reinterpret_cast<const f32x4*>(p.scale + ch) : (f32x4){1.f, 1.f, 1.f, 1.f};
                sf[i] = p.shift ? *reinterpret_cast<const f32x4*>(p.shift + ch) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        // bit j: the pixel of block j lies in image row 0 / H-1, column 0 / W-1
        uint32_t m_top = 0, m_bot = 0, m_lo = 0, m_hi = 0;
#pragma unroll
        for (int j = 0; j < PBW; ++j) {
            const int m = p0 + wave * (16 * PBW) + j * 16 + fr;
            const uint32_t mm = (uint32_t)(m < p.M ? m : 0);
            const uint32_t n = fastdiv(mm, p.div_hw);
            const uint32_t rem = mm - n * (uint32_t)p.HW;
            const uint32_t h = fastdiv(rem, p.div_w);
            const uint32_t w = rem - h * (uint32_t)p.W;
            m_top |= (h == 0u ? 1u : 0u) << j;
            m_bot |= ((int)h == p.H - 1 ? 1u : 0u) << j;
            m_lo |= (w == 0u ? 1u : 0u) << j;
            m_hi |= ((int)w == p.W - 1 ? 1u : 0u) << j;
        }
        // X(tile) landed: the only younger VMEM ops of this wave are the previous tile's 8 stores (a weight reload waits for all)
        if (reloaded) gconv_wait_vmcnt<0>();
        else gconv_wait_vmcnt<4 * PBW>();
        __builtin_amdgcn_s_barrier();                  // ... for every wave; the other buffer is no longer read
        asm volatile("" ::: "memory");
        issue_x(ntile, xb ^ 1, has_next);

        f32x4 acc[4][PBW];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < PBW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const char* xtile = smem + xb * (XR * 128);
#pragma unroll
        for (int ks = 0; ks < 5; ++ks) {
            // this lane's tap of the pair (2 ks, 2 ks + 1); tap 9 does not exist (zero weights): read tap 8 instead
            const int t0 = 2 * ks, t1 = 2 * ks + 1 < 9 ? 2 * ks + 1 : 8;
            const int r0 = t0 / 3, q0 = t0 % 3, r1 = t1 / 3, q1 = t1 % 3;
            const int row = xrow0 + (thalf ? (r1 - 1) * p.W + (q1 - 1) : (r0 - 1) * p.W + (q0 - 1));
            const uint32_t killA = (r0 == 0 ? m_top : 0u) | (r0 == 2 ? m_bot : 0u) | (q0 == 0 ? m_lo : 0u) | (q0 == 2 ? m_hi : 0u);
            const uint32_t killB = (r1 == 0 ? m_top : 0u) | (r1 == 2 ? m_bot : 0u) | (q1 == 0 ? m_lo : 0u) | (q1 == 2 ? m_hi : 0u);
            const uint32_t kill = thalf ? killB : killA;
            const char* xbase = xtile + row * 128;
            const int rsw = row & 7;                   // (+16 j does not change it)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int xsw = ((2 * i + chalf) ^ rsw) << 4;
#pragma unroll
                for (int j = 0; j < PBW; ++j) {
                    frag b = *reinterpret_cast<const frag*>(xbase + j * 2048 + xsw);
                    if ((kill >> j) & 1u) b = (frag){};
                    acc[i][j] = Mma<DT>::run(a[i][ks], b, acc[i][j]);
                }
            }
        }

        // ---- epilogue: MFMA rows 4 fq + e of slab i = channels 64 cb + 16 i + 4 fq + e, pixel 16 j + fr: 8-byte stores ----------
        const int mBase = p0 + wave * (16 * PBW) + fr;
        F16Guard<DT> guard;                            // (rows beyond M were DMA'd as zeros: finite)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ch = cb * 64 + 16 * i + 4 * fq;
#pragma unroll
            for (int j = 0; j < PBW; ++j) {
                const int m = mBase + 16 * j;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e] * sc[i][e] + sf[i][e];
                apply_actn(v, act);
                guard.see(v);
                u32x2 o = {pack2<DT>(v[0], v[1]), pack2<DT>(v[2], v[3])};
                __builtin_amdgcn_raw_buffer_store_b64(o, yrsrc, m < p.M ? (uint32_t)((m * p.C + ch) * 2) : 0x80000000u, 0, 0);
            }
        }
        guard.commit(p.ovf);
        if (!has_next) break;
        tile = ntile;
        xb ^= 1;
    }
    gconv_wait_vmcnt<0>();                             // the look-ahead DMA of the last tile (issued out of range)
#endif  // __HIP_DEVICE_COMPILE__
}

// ---- weight packing: w fp32 [C][cg][3][3] -> [C / 16][5][16][32]; K element k of step ks: tap 2 ks + k / 16, input channel
//      (16 slab + k % 16); zero unless that channel belongs to the row's group (and for the non-existent tap 9) ---------------
template <int DT>
__global__ __launch_bounds__(256) void pack_gconv_kernel(const float* __restrict__ w, void* __restrict__ out, int C, int cg) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)(C / 16) * 5 * 16 * 32;
    if (i >= total) return;
    const int k = (int)(i & 31);
    const int row = (int)((i >> 5) & 15);
    const int ks = (int)((i >> 9) % 5);
    const int slab = (int)((i >> 9) / 5);
    const int tap = 2 * ks + (k >> 4);
    const int o = slab * 16 + row;                     // output channel
    const int c = slab * 16 + (k & 15);                // input channel
    float v = 0.f;
    if (tap < 9 && o / cg == c / cg) v = w[((size_t)o * cg + (c % cg)) * 9 + tap];
    store_elem<DT>(out, (size_t)i, v);
}

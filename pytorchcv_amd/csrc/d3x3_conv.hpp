// d3x3_conv.hpp - dense 3x3 / stride 1 / pad 1 convolution as an implicit GEMM on gfx950 MFMA: the MFMA-bound class of the
// path (ResNet-50: 16 layers, 48 % of the MACs). One 8-wave block per CU, LDS ring of whole K-steps filled by LDS-DMA that
// stays in flight across barriers (counted vmcnt, raw s_barrier), and the two wave groups of a SIMD in opposite phases.
//
// Replaces: nn.Conv2d(3x3, stride 1, padding 1) + nn.BatchNorm2d(eval) + activation of ConvBlock.forward
//           (reference pytorchcv/models/common/conv.py:278-286) at `conv3x3_block` call sites
//           (resnet.py:49,56,120-127 - ResBlock / ResBottleneck.conv2 - vgg.py, preresnet.py), plus the residual add +
//           ReLU of basic-block units (resnet.py:227-228) in the epilogue.
//
// GEMM view (as igemm_conv.hpp): Y^T[ch, pixel] = sum_k Wp[ch, k] X[pixel, k], k = (filter row r, 64-channel slice, filter
// column q) - the packed blob of `plan_conv`'s `conv3` order; A = weights, B = pixels gathered by per-lane DMA source offsets
// (a padded tap is an out-of-range offset: the buffer unit writes zeros). LDS rows are 128 B (one K-step of 64 elements), the
// 16-byte chunk slot s of row r holds K-chunk s ^ (r & 7) (swizzle on the SOURCE side, LDS image lane-linear).
//
// What differs from the generic 4-wave kernel (which reaches ~1000 TFLOP/s per busy block slot and no more):
//   * 512 threads, ONE block per CU: a K-step of a 256-row channel tile costs each wave half the DMA pieces per MFMA.
//   * Ring of NS = 2 or 3 stages; the pieces of stage s + NS - 1 are issued while stage s is computed and are NOT drained at
//     the barriers (raw s_barrier, `s_waitcnt vmcnt(N)` once per stage with N = the younger stage's pieces): every piece has
//     2-3 barrier intervals (1 100+ cycles) to land instead of the one MFMA burst of a __syncthreads() pipeline.
//   * Waves 0-3 and 4-7 (one of each per SIMD) run the same stage one barrier interval apart: while one group issues its
//     fragment reads (and waits for them), the other group's MFMAs own the matrix pipe (MI355X_MICROARCH.md, "Two waves per
//     SIMD"). A stage is four intervals: {reads k-half 0 | MFMA k-half 0 | reads k-half 1 | MFMA k-half 1}.
//   * Tile shapes chosen by the host so that the tile count fills whole rounds of the 256 CUs (pixel tiles of 7 or 13
//     16-pixel blocks per wave: 112 / 208 / 416 / 448 pixels) - the tile-schedule tail of the 128x128 tiling cost 23 %.
//
// Barrier/visibility rules followed (cdna_hip_programming.md, "Read a staged buffer one phase AFTER the wait that retires it"):
//   RAW  every wave waits for ITS pieces of stage s+1 (counted vmcnt) before the barrier that ends stage s; the first read of
//        stage s+1 is issued after that barrier.
//   WAR  every fragment read is retired (lgkmcnt(0)) before the barrier that ends its interval; the slot of stage s-1 is
//        re-filled from the first interval of stage s on, i.e. after both groups' last reads of it.
#pragma once
#include <type_traits>
#include "pcv_common.hpp"
#include "igemm_conv.hpp"     // Mma<DT>, FastDiv

struct D3Params {
    const void* x;          // NHWC [N,H,W,Cin], dense
    const void* w;          // packed weights [rows][Kpad], K = (r, slice, q), rows in MFMA order
    const void* res;        // residual NHWC [M, Cout] or null
    void* y;                // NHWC [M, Ypitch]
    const float* scale;     // [Cout] fp32, never null
    const float* shift;
    uint32_t x_bytes, w_bytes, y_bytes, res_bytes;
    int M;                  // N*H*W
    int Cout, Ypitch;
    int H, W, Cin, HW;
    FastDiv div_hw, div_w;
    int nk;                 // K-steps = 9 * Cin / 64
    int slices;             // Cin / 64
    int Kpad;
    int act, post_act;
    int nChTiles, nTiles;
};

// WC x WP: wave grid (channels x pixels), 8 waves. CBW / PBW: 16-row blocks per wave (channels / pixels).
template <int WC, int WP, int CBW, int PBW> struct D3Cfg {
    static constexpr int BM = 16 * CBW * WC;                 // channel rows per block tile
    static constexpr int BP = 16 * PBW * WP;                 // pixel rows per block tile
    static constexpr int NPA = BM / 8, NPB = BP / 8;         // 1 KB DMA pieces (8 rows x 128 B) per stage
    static constexpr int NPW = (NPA + NPB + 7) / 8;          // pieces per wave per stage (every wave issues exactly this many)
    static constexpr int WL = NPA / 8;                       // ... of which weight pieces
    static constexpr int XL = NPW - WL;                      // ... and activation pieces (pieces past the tile write zeros to pad rows)
    static constexpr int STAGE = NPW * 8 * 1024;             // bytes per ring slot (tile rows + pad rows)
    static constexpr int NS = (3 * STAGE <= 160 * 1024) ? 3 : 2;
    static constexpr int LDS = NS * STAGE;
    static_assert(WC * WP == 8, "eight waves");
    static_assert(NPA % 8 == 0 && CBW % 2 == 0, "weight pieces split evenly over the waves; channel pairs per wave");
    static_assert(2 * STAGE <= 160 * 1024, "two stages must fit the LDS");
};

#if defined(__HIP_DEVICE_COMPILE__)
// The whole persistent loop of one wave group (GRP 0: waves 0-3, GRP 1: waves 4-7, one barrier interval behind). The two
// instantiations are separate straight-line loop nests (no per-interval group branches for the register allocator to join).
template <int DT, int WC, int WP, int CBW, int PBW, int GRP>
__device__ __forceinline__ void d3x3_body(const D3Params& p, char* smem, const int wave) {
    typedef D3Cfg<WC, WP, CBW, PBW> G;
    constexpr int BM = G::BM, BP = G::BP, NPA = G::NPA, NPW = G::NPW, WL = G::WL, XL = G::XL, NS = G::NS;
    constexpr int AHEAD = NS - 1;
    constexpr int H0 = (NPW + 1) / 2;                         // pieces issued in the first half of a stage
    typedef typename Mma<DT>::frag frag;

    const int lane = threadIdx.x & 63;
    const int wc = wave / WP, wp = wave % WP;
    const int lrow = lane >> 3;
    const int cs = (lane & 7) ^ lrow;                         // K-chunk this lane fetches (source-side swizzle)
    const int fr = lane & 15, fq = lane >> 4;

    // ---- this block's tiles: [tile0, tend) of its XCD's contiguous range, stride = blocks per XCD -------------------------
    const int perXcd = (p.nTiles + 7) >> 3;
    const int xcd = blockIdx.x & 7;
    const int tstride = gridDim.x >> 3;                       // host guarantees gridDim.x % 8 == 0
    const int tile0 = xcd * perXcd + (int)(blockIdx.x >> 3);
    const int tend = min(p.nTiles, (xcd + 1) * perXcd);
    if (tile0 >= tend) return;                                 // (whole block: the tile range does not depend on the wave)
    const int nMine = (tend - tile0 + tstride - 1) / tstride;
    const int nk = p.nk;
    const int G_total = nMine * nk;                           // stages this block walks

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);

    // ---- DMA side: the stage being issued (runs AHEAD stages in front of the stage being computed) -----------------------
    int rbase[XL];                 // element offset of tap (0, 0) of this thread's pixel rows
    uint32_t rmask[XL];            // bits 0-2: filter rows inside the image, bits 16-18: filter columns
    uint32_t woff[WL];             // byte offset of this thread's weight rows (chunk cs) in the packed blob
    int ld_tile = tile0, ld_g = 0, ld_k = 0, ld_r = 0, ld_q = 0, ld_cs = 0, ld_slot = 0;

    auto setup = [&](int t) __attribute__((always_inline)) {
        const int chTile = t % p.nChTiles;
        const int tileP0 = (t / p.nChTiles) * BP;
#pragma unroll
        for (int i = 0; i < XL; ++i) {
            const int rt = 8 * (8 * i + wave) + lrow;          // row within the pixel tile
            const int m = tileP0 + rt;
            uint32_t mask = 0;
            int base = 0;
            if (rt < BP && m < p.M) {
                const uint32_t n = fastdiv((uint32_t)m, p.div_hw);
                const uint32_t rem = (uint32_t)m - n * (uint32_t)p.HW;
                const uint32_t ho = fastdiv(rem, p.div_w);
                const uint32_t wo = rem - ho * (uint32_t)p.W;
                base = (((int)n * p.H + (int)ho - 1) * p.W + (int)wo - 1) * p.Cin;
#pragma unroll
                for (int t3 = 0; t3 < 3; ++t3) {
                    mask |= ((uint32_t)((int)ho - 1 + t3) < (uint32_t)p.H ? 1u : 0u) << t3;
                    mask |= ((uint32_t)((int)wo - 1 + t3) < (uint32_t)p.W ? 1u : 0u) << (16 + t3);
                }
            }
            rbase[i] = base;
            rmask[i] = mask;
        }
#pragma unroll
        for (int i = 0; i < WL; ++i) {
            const int wrow = chTile * BM + 8 * (8 * i + wave) + lrow;
            woff[i] = (uint32_t)((wrow * p.Kpad + cs * 8) * 2);          // rows past the blob: out of range -> zeros
        }
    };

    // pieces [I0, I1) of the stage (ld_tile, ld_k) into ring slot ld_slot
    auto dma = [&](auto I0c, auto I1c) __attribute__((always_inline)) {
        constexpr int I0 = decltype(I0c)::value, I1 = decltype(I1c)::value;
        char* sbase = smem + ld_slot * G::STAGE;
        const int koff = (ld_r * p.W + ld_q) * p.Cin + ld_cs * 64 + cs * 8;
#pragma unroll
        for (int i = I0; i < I1; ++i) {
            if (i < WL) {
                char* dst = sbase + (8 * i + wave) * 1024;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, PCV_LDS(dst), 16, woff[i], ld_k * 128, 0, 0);
            } else {
                const int j = i - WL;
                char* dst = sbase + (NPA + 8 * j + wave) * 1024;
                const bool ok = ((rmask[j] >> ld_r) & (rmask[j] >> (16 + ld_q)) & 1u) != 0;
                const uint32_t voff = ok ? (uint32_t)((rbase[j] + koff) * 2) : 0x80000000u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, PCV_LDS(dst), 16, voff, 0, 0, 0);
            }
        }
    };
    auto advance = [&]() __attribute__((always_inline)) {                                    // K order of the blob: (r, slice, q)
        ++ld_g;
        ld_slot = ld_slot + 1 == NS ? 0 : ld_slot + 1;
        if (++ld_k == nk) {
            ld_k = ld_r = ld_q = ld_cs = 0;
            ld_tile += tstride;
            if (ld_tile < tend) setup(ld_tile);
        } else if (++ld_q == 3) {
            ld_q = 0;
            if (++ld_cs == p.slices) { ld_cs = 0; ++ld_r; }
        }
    };
    typedef std::integral_constant<int, 0> C0;
    typedef std::integral_constant<int, H0> CH;
    typedef std::integral_constant<int, NPW> CN;

    // ---- compute side -------------------------------------------------------------------------------------------------------
    f32x4 acc[CBW][PBW];
    frag a[CBW], b[PBW];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < CBW; ++i)
#pragma unroll
            for (int j = 0; j < PBW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    };
    const int afrag = (wc * 16 * CBW + fr) * 128;
    const int bfrag = (BM + wp * 16 * PBW + fr) * 128;
    auto reads = [&](int slot, int kk) __attribute__((always_inline)) {
        const char* sb = smem + slot * G::STAGE;
        const int swz = ((fq + 4 * kk) ^ (fr & 7)) << 4;
#pragma unroll
        for (int i = 0; i < CBW; ++i) a[i] = *reinterpret_cast<const frag*>(sb + afrag + i * 2048 + swz);
#pragma unroll
        for (int j = 0; j < PBW; ++j) b[j] = *reinterpret_cast<const frag*>(sb + bfrag + j * 2048 + swz);
    };
    auto mfmas = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < PBW; ++j)
#pragma unroll
            for (int i = 0; i < CBW; ++i) acc[i][j] = Mma<DT>::run(a[i], b[j], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
    };

    // Epilogue: v = acc * scale + shift -> act -> (+ residual) -> post_act -> one 16-byte NHWC store per (channel pair, pixel
    // block). Branch-free: pad channels / rows past the tile read clamped table entries and an out-of-range (zero) residual and
    // are dropped by the store's range check. Activations: none / ReLU / ReLU6 only (the host sends anything else to the generic
    // kernel) - one inlined copy of this code per wave group.
    // Packed weight row (16 i + rho) of a 64-row group holds channel 32 (i >> 1) + 8 (rho >> 2) + 4 (i & 1) + (rho & 3): lane group
    // fq owns the 8 consecutive channels 32 ip + 8 fq .. + 7 of a pixel (accumulators 2 ip and 2 ip + 1).
    const ActClamp act = make_act(p.act), pact = make_act(p.post_act);
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res), 0, p.res != nullptr ? p.res_bytes : 0u, 0x00020000);
    auto epilogue = [&](int t) __attribute__((always_inline)) {
        const int chTile = t % p.nChTiles;
        const int tileP0 = (t / p.nChTiles) * BP;
        const int mBase = tileP0 + wp * 16 * PBW + fr;
#pragma unroll
        for (int ip = 0; ip < CBW / 2; ++ip) {
            const int ch0 = chTile * BM + wc * 16 * CBW + 32 * ip + 8 * fq;
            const bool chok = ch0 < p.Cout;
            const int chl = chok ? ch0 : 0;                      // table index of a pad channel: any valid one (never stored)
            const f32x4 s0 = *reinterpret_cast<const f32x4*>(p.scale + chl), s1 = *reinterpret_cast<const f32x4*>(p.scale + chl + 4);
            const f32x4 h0 = *reinterpret_cast<const f32x4*>(p.shift + chl), h1 = *reinterpret_cast<const f32x4*>(p.shift + chl + 4);
            u32x4 rr[PBW];
#pragma unroll
            for (int j = 0; j < PBW; ++j) {
                const int m = mBase + 16 * j;
                const uint32_t roff = (chok && m < p.M) ? (uint32_t)(((size_t)m * p.Cout + ch0) * 2) : 0x80000000u;
                rr[j] = __builtin_amdgcn_raw_buffer_load_b128(rrsrc, roff, 0, 0);        // no residual: zero records -> zeros
            }
#pragma unroll
            for (int j = 0; j < PBW; ++j) {
                const int m = mBase + 16 * j;
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[2 * ip][j][e] * s0[e] + h0[e];
                    v[4 + e] = acc[2 * ip + 1][j][e] * s1[e] + h1[e];
                }
                clampn<8>(v, act);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float lo, hi;
                    unpack2<DT>(rr[j][e], lo, hi);
                    v[2 * e] += lo;
                    v[2 * e + 1] += hi;
                }
                clampn<8>(v, pact);
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = pack2<DT>(v[2 * e], v[2 * e + 1]);
                const bool ok = chok && m < p.M;
                const uint32_t boff = ok ? (uint32_t)(((size_t)m * p.Ypitch + ch0) * 2) : 0x80000000u;
                __builtin_amdgcn_raw_buffer_store_b128(o, yrsrc, boff, 0, 0);
            }
        }
    };

    // ---- prologue: the first AHEAD stages in flight, stage 0 landed -----------------------------------------------------------
    setup(tile0);
    zero_acc();
#pragma unroll
    for (int s = 0; s < AHEAD; ++s) {
        if (ld_g < G_total) {
            dma(C0{}, CN{});
            advance();
        }
    }
    if (AHEAD == 2 && G_total > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    int rd_slot = 0, k = 0, cur_tile = tile0, ep_tile = tile0;
    bool ep = false;
    auto sync = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int s = 0;; ++s) {
        const bool more = ld_g < G_total;                     // a stage s + AHEAD exists and is issued during this stage
        // ---- interval 0: group 0 reads k-half 0 of stage s | group 1 finishes stage s - 1 --------------------------------------
        // (s == G_total: the tail - group 1's last k-half and both groups' last epilogue)
        if constexpr (GRP == 1) {
            if (more) dma(C0{}, CH{});
            if (s > 0) mfmas();
        }
        if (ep) {
            epilogue(ep_tile);
            zero_acc();
        }
        if (s == G_total) break;
        if constexpr (GRP == 0) {
            reads(rd_slot, 0);
            if (more) dma(C0{}, CH{});
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        sync();
        // ---- interval 1 ------------------------------------------------------------------------------------------------------------
        if constexpr (GRP == 1) reads(rd_slot, 0);
        if (more) {
            dma(CH{}, CN{});
            advance();
        }
        if constexpr (GRP == 0) mfmas();
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        sync();
        // ---- interval 2 ------------------------------------------------------------------------------------------------------------
        if constexpr (GRP == 0) {
            reads(rd_slot, 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
            mfmas();
        }
        sync();
        // ---- interval 3: stage s + 1 must have landed when it ends -----------------------------------------------------------------
        if constexpr (GRP == 0) {
            mfmas();
        } else {
            reads(rd_slot, 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        ep = (k == nk - 1);
        ep_tile = cur_tile;
        if (++k == nk) {
            k = 0;
            cur_tile += tstride;
        }
        rd_slot = rd_slot + 1 == NS ? 0 : rd_slot + 1;
        if (AHEAD == 2 && more) {
            // stage s + 2 was issued during this stage: its NPW pieces may stay in flight, everything older has landed
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        sync();
    }
}
#endif  // __HIP_DEVICE_COMPILE__

template <int DT, int WC, int WP, int CBW, int PBW>
__global__ __launch_bounds__(512, 2) void d3x3_kernel(const D3Params p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // waves w and w + 4 share a SIMD (a workgroup's waves are dealt to the SIMDs cyclically): one wave of each group per SIMD
    if (wave < 4) d3x3_body<DT, WC, WP, CBW, PBW, 0>(p, smem, wave);
    else d3x3_body<DT, WC, WP, CBW, PBW, 1>(p, smem, wave);
#endif  // __HIP_DEVICE_COMPILE__
}

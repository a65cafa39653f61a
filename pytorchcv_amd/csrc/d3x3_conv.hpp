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
// Structure (what differs from the generic 4-wave kernel, which reaches ~1000 TFLOP/s per busy block slot and no more):
//   * 512 threads, ONE block per CU, ring of NS = 2 or 3 whole K-steps. Waves 0-3 ("group 0") and 4-7 ("group 1") - one of
//     each per SIMD - run the same stage one barrier interval apart: while one group issues its fragment reads and waits for
//     them, the other group's MFMAs own the matrix pipe (MI355X_MICROARCH.md, "Two waves per SIMD"). With KS = 2 a stage is
//     two intervals {reads of the whole K-step | its MFMAs}, with KS = 1 four {reads k-half 0 | MFMA | reads k-half 1 | MFMA}.
//   * ALL LDS-DMA is issued by group 0, behind its fragment reads, in the interval where group 1 computes: a wave issues in
//     order, so a DMA piece in front of an MFMA burst delays the burst by the time the texture path takes to accept the
//     piece (1 KB at 64 B/clk/CU, 8 waves queueing: measured 100+ cycles per piece, in-kernel stamps) - in the reading group
//     that time is hidden behind the other group's MFMAs. The pieces of stage s + NS - 1 are issued during stage s and are
//     NOT drained at the barriers (raw s_barrier; `s_waitcnt vmcnt(N)` once per stage, N = the younger stage's pieces).
//   * Tile shapes chosen by the host so that the tile count fills whole rounds of the 256 CUs (pixel tiles of 7 or 13
//     16-pixel blocks per wave: 112 / 208 / 224 / 416 / 448 pixels) - the tile-schedule tail of the 128x128 tiling cost 23 %.
//
// Barrier/visibility rules followed (cdna_hip_programming.md, "Read a staged buffer one phase AFTER the wait that retires it"):
//   RAW  group 0 waits for the pieces of stage s + 1 (counted vmcnt) before the barrier that ends stage s; the first read of
//        stage s + 1 is issued after that barrier.
//   WAR  every fragment read is retired (lgkmcnt(0)) before the barrier that ends its interval; the slot of stage s - 1 is
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
    uint32_t* dbg;          // diagnostic builds only (-DD3X3_STAMPS): per-wave s_memtime stamps of one block, 8 x 64 dwords
};

// In-kernel stamps (cdna_hip_programming.md section 7): a diagnostic build (-DD3X3_STAMPS) times ONE section per stage - the
// stamp that opens it and the stamp that closes it, s_memtime low words written into the lanes of one VGPR per wave - and
// rotates the section from stage to stage (stages 4..30: every section three times), so a stage carries two s_memtime round
// trips instead of nine. Stamp points sit behind a barrier or an explicit lgkmcnt(0), where no LDS read is outstanding
// (s_memtime returns through lgkmcnt). The product build compiles none of this.
#ifdef D3X3_STAMPS
#define D3_STAMP(slot)                                                                                     \
    do {                                                                                                   \
        const int q__ = s - 4;                                                                             \
        if (q__ >= -1 && q__ < 27) {                                                                       \
            const int sel__ = (q__ + 9) % 9;                                                               \
            int l__ = -1;                                                                                  \
            if (q__ >= 0 && (slot) == sel__) l__ = 2 * q__ + 1;                                            \
            else if (q__ >= 0 && sel__ > 0 && (slot) == sel__ - 1) l__ = 2 * q__;                          \
            else if ((slot) == 8 && (q__ + 1) % 9 == 0 && q__ + 1 < 27) l__ = 2 * (q__ + 1);               \
            if (l__ >= 0) {                                                                                \
                const uint64_t t__ = __builtin_amdgcn_s_memtime();                                         \
                const int v__ = __builtin_amdgcn_readfirstlane((int)(uint32_t)t__);                        \
                const int i__ = __builtin_amdgcn_readfirstlane(l__);                                       \
                asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(stamps) : "s"(v__), "s"(i__) : "m0"); \
            }                                                                                              \
        }                                                                                                  \
    } while (0)
#else
#define D3_STAMP(slot) do { } while (0)
#endif

// WC x WP: wave grid (channels x pixels), 8 waves. CBW / PBW: 16-row blocks per wave (channels / pixels).
// KS: K-halves (32 elements each) per read / MFMA section.
template <int WC, int WP, int CBW, int PBW, int KS> struct D3Cfg {
    static constexpr int BM = 16 * CBW * WC;                 // channel rows per block tile
    static constexpr int BP = 16 * PBW * WP;                 // pixel rows per block tile
    static constexpr int NPA = BM / 8, NPB = BP / 8;         // 1 KB DMA pieces (8 rows x 128 B) per stage
    static constexpr int NPL = (NPA + NPB + 3) / 4;          // pieces per LOADING wave (waves 0-3) per stage
    static constexpr int WL = NPA / 4;                       // ... of which weight pieces
    static constexpr int XL = NPL - WL;                      // ... and activation pieces (pieces past the tile write zeros to pad rows)
    static constexpr int STAGE = NPL * 4 * 1024;             // bytes per ring slot (tile rows + pad rows)
    static constexpr int NS = (3 * STAGE <= 160 * 1024) ? 3 : 2;
    static constexpr int LDS = NS * STAGE;
    static_assert(WC * WP == 8, "eight waves");
    static_assert(KS == 1 || KS == 2, "one or two K-halves per section");
    static_assert(NPA % 4 == 0 && CBW % 2 == 0, "weight pieces split evenly over the loading waves; channel pairs per wave");
    static_assert(2 * STAGE <= 160 * 1024, "two stages must fit the LDS");
    static_assert(XL <= 15, "filter-row/column masks of the activation pieces are packed 5 per register, three registers");
};

#if defined(__HIP_DEVICE_COMPILE__)
// The whole persistent loop of one wave group (GRP 0: waves 0-3, GRP 1: waves 4-7, one barrier interval behind). The two
// instantiations are separate straight-line loop nests (no per-interval group branches for the register allocator to join).
template <int DT, int WC, int WP, int CBW, int PBW, int KS, int GRP>
__device__ __forceinline__ void d3x3_body(const D3Params& p, char* smem, const int wave) {
    typedef D3Cfg<WC, WP, CBW, PBW, KS> G;
    constexpr int BM = G::BM, BP = G::BP, NPA = G::NPA, NPL = G::NPL, WL = G::WL, XL = G::XL, NS = G::NS;
    constexpr int AHEAD = NS - 1;
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

    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res), 0, p.res != nullptr ? p.res_bytes : 0u, 0x00020000);

    // ---- DMA side (group 0 only): the stage being issued runs AHEAD stages in front of the stage being computed -----------
    // Loading wave w (0..3) owns pieces 4 i + w, i < NPL: the first WL are weight rows, the rest pixel rows.
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);
    int rbase[XL];                 // element offset of tap (0, 0) of this thread's pixel rows
    uint32_t rmask[3] = {0u, 0u, 0u};  // 6 bits per pixel row (3 filter rows, 3 filter columns inside the image), 5 rows per register
    uint32_t woff0 = 0;            // byte offset of this thread's first weight row (chunk cs) in the packed blob
    int ld_tile = tile0, ld_g = 0, ld_k = 0, ld_r = 0, ld_q = 0, ld_cs = 0, ld_slot = 0;

    auto setup = [&](int t) __attribute__((always_inline)) {
        const int chTile = t % p.nChTiles;
        const int tileP0 = (t / p.nChTiles) * BP;
        rmask[0] = rmask[1] = rmask[2] = 0u;
#pragma unroll
        for (int i = 0; i < XL; ++i) {
            const int rt = 8 * (4 * i + wave) + lrow;          // row within the pixel tile
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
                    mask |= ((uint32_t)((int)wo - 1 + t3) < (uint32_t)p.W ? 1u : 0u) << (3 + t3);
                }
            }
            rbase[i] = base;
            rmask[i / 5] |= mask << (6 * (i % 5));
        }
        woff0 = (uint32_t)(((chTile * BM + 8 * wave + lrow) * p.Kpad + cs * 8) * 2);     // rows past the blob: out of range -> zeros
    };

    // all NPL pieces of the stage (ld_tile, ld_k) into ring slot ld_slot, then step to the next stage
    auto dma_stage = [&]() __attribute__((always_inline)) {
        char* sbase = smem + ld_slot * G::STAGE;
        const int koff = (ld_r * p.W + ld_q) * p.Cin + ld_cs * 64 + cs * 8;
        const uint32_t wstep = (uint32_t)(32 * p.Kpad * 2);                               // 4 pieces x 8 rows further down the blob
#pragma unroll
        for (int i = 0; i < WL; ++i) {
            char* dst = sbase + (4 * i + wave) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, PCV_LDS(dst), 16, woff0 + i * wstep, ld_k * 128, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < XL; ++j) {
            char* dst = sbase + (NPA + 4 * j + wave) * 1024;
            const uint32_t m6 = rmask[j / 5] >> (6 * (j % 5));
            const bool ok = ((m6 >> ld_r) & (m6 >> (3 + ld_q)) & 1u) != 0;
            const uint32_t voff = ok ? (uint32_t)((rbase[j] + koff) * 2) : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, PCV_LDS(dst), 16, voff, 0, 0, 0);
        }
        ++ld_g;                                                // K order of the blob: (r, slice, q)
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

    // ---- compute side -------------------------------------------------------------------------------------------------------
    f32x4 acc[CBW][PBW];
    frag a[KS][CBW], b[KS][PBW];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < CBW; ++i)
#pragma unroll
            for (int j = 0; j < PBW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    };
    const int afrag = (wc * 16 * CBW + fr) * 128;
    const int bfrag = (BM + wp * 16 * PBW + fr) * 128;
    // section h (of 2 / KS) of the stage in `slot`: K-halves h * KS .. h * KS + KS - 1
    auto reads = [&](int slot, int h) __attribute__((always_inline)) {
        const char* sb = smem + slot * G::STAGE;
#pragma unroll
        for (int u = 0; u < KS; ++u) {
            const int swz = ((fq + 4 * (h * KS + u)) ^ (fr & 7)) << 4;
#pragma unroll
            for (int i = 0; i < CBW; ++i) a[u][i] = *reinterpret_cast<const frag*>(sb + afrag + i * 2048 + swz);
#pragma unroll
            for (int j = 0; j < PBW; ++j) b[u][j] = *reinterpret_cast<const frag*>(sb + bfrag + j * 2048 + swz);
        }
    };
    auto mfmas = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int u = 0; u < KS; ++u)
#pragma unroll
            for (int j = 0; j < PBW; ++j)
#pragma unroll
                for (int i = 0; i < CBW; ++i) acc[i][j] = Mma<DT>::run(a[u][i], b[u][j], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
    };
    // every fragment read of this wave has returned (a compiler-visible wait: lgkmcnt(0), the other counters untouched)
    auto reads_done = [&]() __attribute__((always_inline)) { __builtin_amdgcn_s_waitcnt(0xC07F); };

    // Epilogue: v = acc * scale + shift -> act -> (+ residual) -> post_act -> one 16-byte NHWC store per (channel pair, pixel
    // block). Branch-free: pad channels / rows past the tile read clamped table entries and an out-of-range (zero) residual and
    // are dropped by the store's range check. Activations: none / ReLU / ReLU6 only (the host sends anything else to the generic
    // kernel) - one inlined copy of this code per wave group.
    // Packed weight row (16 i + rho) of a 64-row group holds channel 32 (i >> 1) + 8 (rho >> 2) + 4 (i & 1) + (rho & 3): lane group
    // fq owns the 8 consecutive channels 32 ip + 8 fq .. + 7 of a pixel (accumulators 2 ip and 2 ip + 1).
    const ActClamp act = make_act(p.act), pact = make_act(p.post_act);
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

    auto sync = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- prologue: group 0 puts the first AHEAD stages in flight and waits for stage 0 -----------------------------------------
    zero_acc();
    if constexpr (GRP == 0) {
        setup(tile0);
#pragma unroll
        for (int s = 0; s < AHEAD; ++s)
            if (ld_g < G_total) dma_stage();
        if (AHEAD == 2 && G_total > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPL) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    sync();

    int rd_slot = 0, k = 0, cur_tile = tile0, ep_tile = tile0;
    bool ep = false;
#ifdef D3X3_STAMPS
    int stamps = 0;
#endif
    for (int s = 0;; ++s) {
        // ---- interval 0: group 0 reads stage s (first section) and issues the DMA of stage s + AHEAD | group 1 finishes stage
        // s - 1. (s == G_total: the tail - group 1's last section and both groups' last epilogue)
        if constexpr (GRP == 1) {
            if (s > 0) mfmas();
        }
        if (ep) {
            epilogue(ep_tile);
            zero_acc();
        }
        if (s == G_total) break;
        bool more = false;
        if constexpr (GRP == 0) {
            reads(rd_slot, 0);
            more = ld_g < G_total;
            if (more) dma_stage();
            reads_done();
        }
        D3_STAMP(0);
        sync();
        D3_STAMP(1);
        // ---- interval 1 ------------------------------------------------------------------------------------------------------------
        if constexpr (GRP == 0) {
            mfmas();
        } else {
            reads(rd_slot, 0);
            reads_done();
        }
        if constexpr (KS == 1) {
            D3_STAMP(2);
            sync();
            D3_STAMP(3);
            // ---- interval 2 --------------------------------------------------------------------------------------------------------
            if constexpr (GRP == 0) {
                reads(rd_slot, 1);
                reads_done();
            } else {
                mfmas();
            }
            D3_STAMP(4);
            sync();
            D3_STAMP(5);
            // ---- interval 3 --------------------------------------------------------------------------------------------------------
            if constexpr (GRP == 0) {
                mfmas();
            } else {
                reads(rd_slot, 1);
                reads_done();
            }
        }
        ep = (k == nk - 1);
        ep_tile = cur_tile;
        if (++k == nk) {
            k = 0;
            cur_tile += tstride;
        }
        rd_slot = rd_slot + 1 == NS ? 0 : rd_slot + 1;
        D3_STAMP(6);
        // ---- end of the stage: stage s + 1 has landed (group 0 issued it one or two stages ago) -----------------------------------
        if constexpr (GRP == 0) {
            if (AHEAD == 2 && more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPL) : "memory");     // stage s + 2 may stay in flight
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        D3_STAMP(7);
        sync();
        D3_STAMP(8);
    }
#ifdef D3X3_STAMPS
    if (p.dbg != nullptr && blockIdx.x == 16) p.dbg[wave * 64 + lane] = (uint32_t)stamps;
#endif
}
#endif  // __HIP_DEVICE_COMPILE__

template <int DT, int WC, int WP, int CBW, int PBW, int KS>
__global__ __launch_bounds__(512, 2) void d3x3_kernel(const D3Params p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // waves w and w + 4 share a SIMD (a workgroup's waves are dealt to the SIMDs cyclically): one wave of each group per SIMD
    if (wave < 4) d3x3_body<DT, WC, WP, CBW, PBW, KS, 0>(p, smem, wave);
    else d3x3_body<DT, WC, WP, CBW, PBW, KS, 1>(p, smem, wave);
#endif  // __HIP_DEVICE_COMPILE__
}

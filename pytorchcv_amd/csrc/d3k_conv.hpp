// d3k_conv.hpp - dense 3x3 / stride 1 / pad 1 convolution with 128 INPUT channels on 28-pixel-wide maps (ResNet stage 2: four layers of
// ResNet-50, three or four of ResNet-18 / 34), gfx950 MFMA: the d3c recipe (d3c_conv.hpp) with the weights of a 128-channel tile - 128 x 1 152
// = 295 KB - in the registers of four waves, most of them in AGPRs.
//
// Replaces: nn.Conv2d(128 -> Cout, 3x3, stride 1, padding 1) + nn.BatchNorm2d(eval) + activation of ConvBlock.forward (reference
//           pytorchcv/models/common/conv.py:278-286) at ResBottleneck.conv2 / ResBlock.conv1, conv2 of the 28 x 28 stage (resnet.py:49,56,
//           120-127), plus the residual add + ReLU of basic-block units (resnet.py:227-228) in the epilogue. Same K order (filter row, 64-channel
//           slice, filter column), same MFMA chain per accumulator, same epilogue arithmetic as d3q / d3w / igemm: bit-identical results.
//
// Why (round 4). On d3w's 128 x 448 tile these layers pull ~107 KB per filter-row step (58 KB of activations + 49 KB of weights) through
// the ~16 B/clk LDS-DMA path for 5 376 MFMA cycles: 68 us per layer at batch 256 against ~30 us of MFMA time. Here:
//   * a block owns 128 output channels; wave w holds the 32 x 1 152 weights of channels 32 w .. as 72 MFMA A fragments = 288 registers:
//     the first 64 in the 256 AGPRs, the last 8 in VGPRs. The compiler's own MFMA selection takes A from VGPRs only (left to it, AGPR-parked
//     weights are copied back before every use), so the MFMAs are inline asm with the A operand constrained to "a" / "v";
//   * a tile = 4 output rows of one image x 128 channels (112 pixels = 7 pixel blocks, every wave computes all seven for its channels);
//     its 6 input rows x 2 channel slices are staged ONCE by LDS-DMA (43 KB for all nine taps: 5 B/clk) into an image with a pitch of 32 rows
//     per image row whose rows 0 and 29..31 are zero: taps are row offsets, rows outside the image arrive as zeros through a per-image
//     descriptor, no masks, no barrier inside a tile; 1 792 tiles at batch 256 = exactly seven rounds of 256 CUs;
//   * pixel-block-outer K loop with the finished block's epilogue between the next block's MFMAs (d3c_conv.hpp).
// LDS image: [slot][slice][staged row][32 rows x 128 B]; LDS row u of an image row holds pixel u - 1; 16-byte chunk slot s of LDS row R holds
// K-chunk s ^ (R & 7).
#pragma once
#include <type_traits>
#include "pcv_common.hpp"
#include "igemm_conv.hpp"     // Mma<DT>
#include "d3q_conv.hpp"       // D3Params, D3Tiles, d3q_tiles

struct D3KCfg {
    static constexpr int THREADS = 256;
    static constexpr int W = 28, ROWS = 4;                   // map width; output rows per tile
    static constexpr int CIN = 128, SLICES = 2;
    static constexpr int BM = 128, BP = ROWS * W;            // 128 channels x 112 pixels
    static constexpr int NBLK = BP / 16;                     // 7 pixel blocks (every wave: all of them)
    static constexpr int PITCH = 32;                         // LDS rows per staged image row
    static constexpr int SROWS = ROWS + 2;                   // staged image rows
    static constexpr int ROWB = PITCH * 128;                 // 4 096 B: one staged row of one slice
    static constexpr int SLICEB = SROWS * ROWB;              // 24 576 B
    static constexpr int SLOT = SLICES * SLICEB;             // 49 152 B
    static constexpr int DUMP = 2 * SLOT;                    // (the row behind slot 1's last piece: written with zeros)
    static constexpr int LDS = DUMP + 1024;
    static constexpr int PPW = SROWS * SLICES;               // pieces per wave per tile: 12 (wave w: pixels 8 w .. 8 w + 7 of every staged row and slice)
    static constexpr int KH = 9 * SLICES * 2;                // 36 K-halves
    static constexpr int KHA = 32;                           // ... of which the first 32 live in AGPRs (64 fragments = 256 registers)
};

#if defined(__HIP_DEVICE_COMPILE__)
// Wait states between an inline-asm MFMA and the first vector instruction that touches its accumulator, inserted by hand because the
// compiler pads nothing for an asm statement (cdna_hip_programming.md section 5.7). -DD3K_DROP_HAZARD_PADS (tests/test_isa_hazards.py
// ONLY) removes them, to prove that the checker notices.
#ifdef D3K_DROP_HAZARD_PADS
#define D3K_MFMA_PAD ""
#else
#define D3K_MFMA_PAD "s_nop 11"
#endif

// c = a . b (+ c) with the A fragment in AGPRs (AG) or VGPRs. No hazard the compiler would have had to pad: B comes from an LDS read (a
// register dependency it tracks), an accumulator is read by vector instructions a whole pixel block later.
template <int DT, bool AG, bool ZERO>
__device__ __forceinline__ void d3k_mma(f32x4& c, const typename Mma<DT>::frag& a, const typename Mma<DT>::frag& b) {
    if constexpr (DT == PCV_BF16) {
        if constexpr (AG && ZERO) asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=v"(c) : "a"(a), "v"(b));
        else if constexpr (AG) asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "a"(a), "v"(b));
        else if constexpr (ZERO) asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=v"(c) : "v"(a), "v"(b));
        else asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    } else {
        if constexpr (AG && ZERO) asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=v"(c) : "a"(a), "v"(b));
        else if constexpr (AG) asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "a"(a), "v"(b));
        else if constexpr (ZERO) asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=v"(c) : "v"(a), "v"(b));
        else asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    }
}

template <int DT>
__device__ __forceinline__ void d3k_body(const D3Params& p, char* smem) {
    typedef D3KCfg G;
    typedef typename Mma<DT>::frag frag;
    typedef __attribute__((address_space(3))) char lds_char;
    typedef const __attribute__((address_space(3))) frag* lds_fptr;
    constexpr int KH = G::KH, KHA = G::KHA, NB = G::NBLK;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // channels 32 wave .. of the tile; pixels 8 wave .. of every staged row (DMA)
    const int fr = lane & 15, fq = lane >> 4;
    const int lrow = lane >> 3;
    const uint32_t lds0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(size_t)PCV_LDS(smem));
    const D3Tiles T = d3q_tiles(p);
    if (T.nMine == 0) return;

    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    const bool has_res = p.res != nullptr;
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res), 0, has_res ? p.res_bytes : 0u, 0x00020000);
    const int tilesPerImage = (p.H + G::ROWS - 1) / G::ROWS;

    // ---- LDS row 0 of every staged row (pixel -1: the left padding) in both slots; the DMA never writes it. Rows 29..31 (and row 0 of the
    // next staged row again) are written with zeros by the fourth piece of every row, whose pixels 28..31 do not exist. ----
    for (int i = threadIdx.x; i < 2 * G::SLICES * G::SROWS * 8; i += G::THREADS) {     // 16-byte chunks
        const int chunk = i & 7, rr = i >> 3;                                            // rr: (slot, slice, staged row)
        *reinterpret_cast<__attribute__((address_space(3))) u32x4*>((size_t)(lds0 + (uint32_t)(rr * G::ROWB + chunk * 16))) = (u32x4){0u, 0u, 0u, 0u};
    }

    // ---- weights: A fragments of this wave's 32 channels, all 36 K-halves (K order: filter row, slice, filter column) ----
    frag A[KH][2];
    f32x4 es0, es1, eh0, eh1;
    auto load_weights = [&](int chTile) __attribute__((always_inline)) {
#pragma unroll
        for (int kh = 0; kh < KH; ++kh)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const uint32_t row = (uint32_t)(chTile * G::BM + wave * 32 + i * 16 + fr);
                const uint32_t off = (row * (uint32_t)p.Kpad + (uint32_t)((kh >> 1) * 64 + (fq + 4 * (kh & 1)) * 8)) * 2u;     // rows past the blob: zeros
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, off, 0, 0);
                A[kh][i] = __builtin_bit_cast(frag, v);
            }
        const int ch0 = chTile * G::BM + wave * 32 + 8 * fq;
        const int chl = ch0 < p.Cout ? ch0 : 0;                  // pad channels: any valid entry (never stored)
        es0 = *reinterpret_cast<const f32x4*>(p.scale + chl); es1 = *reinterpret_cast<const f32x4*>(p.scale + chl + 4);
        eh0 = *reinterpret_cast<const f32x4*>(p.shift + chl); eh1 = *reinterpret_cast<const f32x4*>(p.shift + chl + 4);
    };

    // ---- fragment addresses: block j, filter column q (row offset q, swizzle key (px + q) & 7); filter row r = + r * 4096, slice = + 24576 ----
    uint32_t ba[NB][3];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int pix = 16 * j + fr;                             // pixel inside the tile
        const int py = pix / G::W, px = pix - py * G::W;
#pragma unroll
        for (int q = 0; q < 3; ++q)
            ba[j][q] = lds0 + (uint32_t)((py * G::PITCH + px + q) * 128 + ((fq ^ ((px + q) & 7)) << 4));
    }

    // ---- DMA: wave w stages pixels 8 w .. 8 w + 7 (LDS rows 1 + 8 w ..) of every (staged row ry, slice): piece I = 2 ry + slice, I = 0 .. 11.
    // Source: a descriptor over the tile's IMAGE only; per-lane offset = pixel and K-chunk of this lane + (y0 - 1 + ry) image rows + the slice:
    // a row above or below the image is out of range and arrives as zeros, and so do pixels 28..31 of wave 3's pieces (offset 2^31). ----
    const uint32_t imgBytes = (uint32_t)(p.H * G::W * G::CIN * 2);
    const uint32_t lanesrc = (wave == 3 && lrow >= 4) ? 0x80000000u
                                                      : (uint32_t)((8 * wave + lrow) * (G::CIN * 2) + (((lane & 7) ^ ((1 + lrow) & 7)) << 4));
    uint32_t vt = 0x80000000u;                                    // the next tile's offset of piece 0 (2^31: no next tile - zeros into the free slot)
    __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, 0, 0x00020000);
    auto dma_setup = [&](bool more, int n, int y0) __attribute__((always_inline)) {
        xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.x)) + (size_t)n * imgBytes, 0, imgBytes, 0x00020000);
        vt = (more && lanesrc != 0x80000000u) ? lanesrc + (uint32_t)((y0 - 1) * (G::W * G::CIN * 2)) : 0x80000000u;
    };
    auto dma_piece = [&](auto Ic, int slot) __attribute__((always_inline)) {
        constexpr int I = decltype(Ic)::value, ry = I >> 1, sl = I & 1;
        const uint32_t dst = lds0 + (uint32_t)(slot * G::SLOT + sl * G::SLICEB + ry * G::ROWB + 128 + wave * 1024);
        // (a row above the image: y0 - 1 + ry = -1 wraps below zero = beyond num_records; the 2^31 marker stays beyond it under these adds)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_char*)(size_t)dst, 16, vt + (uint32_t)(ry * (G::W * G::CIN * 2) + sl * 128), 0, 0, 0);
    };
    auto tile_pos = [&](int t, int& chTile, int& n, int& y0) __attribute__((always_inline)) {
        chTile = t % p.nChTiles;
        const int pt = t / p.nChTiles;
        n = pt / tilesPerImage;
        y0 = (pt - n * tilesPerImage) * G::ROWS;
    };

    // K loop order: PIXEL BLOCK outer, K-half inner (d3c_conv.hpp): a block's two accumulators run through all 36 K-halves (72 MFMAs), then
    // the block is finished and stored while the next block's MFMAs run. One fragment read per step, RD steps ahead.
    constexpr int RD = 5, RING = 8, NSTEP = NB * KH;
    f32x4 cacc[2][2];                                             // [block parity][channel half]
    frag bq[RING];
    uint32_t bs[NB][3];                                           // ba + this tile's slot (set per tile)
    auto rd = [&](auto STc) __attribute__((always_inline)) {
        constexpr int st = decltype(STc)::value;
        constexpr int j = st / KH, kh = st - KH * j, ks = kh >> 1, r = ks / 6, sl = (ks / 3) % 2, q = ks % 3, h = kh & 1;
        const uint32_t a = bs[j][q] ^ (uint32_t)(h << 6);                           // (kc + 4) ^ key = (kc ^ key) ^ 4: bit 6 of the address
        bq[st % RING] = *reinterpret_cast<lds_fptr>((size_t)a + (size_t)(r * G::ROWB + sl * G::SLICEB));
    };
    // activations none / ReLU / ReLU6 as branch-free clamps to launch-uniform bounds (d3c_conv.hpp)
    const float alo = (p.act == PCV_ACT_RELU || p.act == PCV_ACT_RELU6) ? 0.f : -INFINITY, ahi = p.act == PCV_ACT_RELU6 ? 6.f : INFINITY;
    const float plo = (p.post_act == PCV_ACT_RELU || p.post_act == PCV_ACT_RELU6) ? 0.f : -INFINITY, phi = p.post_act == PCV_ACT_RELU6 ? 6.f : INFINITY;
    const float clo = alo > plo ? alo : plo, chi = ahi < phi ? ahi : phi;
    u32x4 rrq[NB];                                                // the tile's residual pieces (requested at the tile's start)
    u32x4 opend;                                                  // the packed outputs of the block being finished
    auto epi_part = [&](auto HRc, auto Jc, auto Ec, F16Guard<DT>& guard) __attribute__((always_inline)) {
        constexpr bool HR = decltype(HRc)::value;
        constexpr int j = decltype(Jc)::value, e = decltype(Ec)::value;          // output dword e: values 2 e, 2 e + 1 of the lane's 8 channels
        constexpr int half = e >> 1, k0 = 2 * (e & 1);                                // accumulator (channel half), element pair
        const f32x4& sc = half == 0 ? es0 : es1;
        const f32x4& sh = half == 0 ? eh0 : eh1;
        float v0 = cacc[j & 1][half][k0] * sc[k0] + sh[k0], v1 = cacc[j & 1][half][k0 + 1] * sc[k0 + 1] + sh[k0 + 1];
        if constexpr (HR) {
            v0 = __builtin_elementwise_minimum(__builtin_elementwise_maximum(v0, alo), ahi);
            v1 = __builtin_elementwise_minimum(__builtin_elementwise_maximum(v1, alo), ahi);
            float lo, hi;
            unpack2<DT>(rrq[j][e], lo, hi);
            v0 += lo;
            v1 += hi;
            v0 = __builtin_elementwise_minimum(__builtin_elementwise_maximum(v0, plo), phi);
            v1 = __builtin_elementwise_minimum(__builtin_elementwise_maximum(v1, plo), phi);
        } else {
            v0 = __builtin_elementwise_minimum(__builtin_elementwise_maximum(v0, clo), chi);
            v1 = __builtin_elementwise_minimum(__builtin_elementwise_maximum(v1, clo), chi);
        }
        guard.see2(v0, v1);
        opend[e] = pack2<DT>(v0, v1);
    };
    auto epi_store = [&](int j, int ch0, int mTile, int mEnd) __attribute__((always_inline)) {
        const int m = mTile + 16 * j + fr;
        const uint32_t boff = (ch0 < p.Cout && m < mEnd) ? (uint32_t)((m * p.Ypitch + ch0) * 2) : 0x80000000u;     // (the host keeps y below 2 GiB)
        __builtin_amdgcn_raw_buffer_store_b128(opend, yrsrc, boff, 0, 0);
    };
    // one tile: 252 steps; the 12 pieces of the NEXT tile's patch go out every 20 steps from step 4 (into the other slot: every wave left it
    // before the barrier that ended the last tile)
    auto tile_steps = [&](auto HRc, int slot, int ch0, int mTile, int mEnd) __attribute__((always_inline)) {
        constexpr bool HR = decltype(HRc)::value;
        F16Guard<DT> guard;
        if constexpr (HR) {
            // the skip tensor's pieces of the whole tile, in FRONT of this tile's LDS-DMA pieces (vmcnt retires in order)
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int m = mTile + 16 * j + fr;
                const uint32_t roff = (ch0 < p.Cout && m < mEnd) ? (uint32_t)((m * p.Cout + ch0) * 2) : 0x80000000u;
                rrq[j] = __builtin_amdgcn_raw_buffer_load_b128(rrsrc, roff, 0, 0);
            }
        }
        rd(std::integral_constant<int, 0>{}); rd(std::integral_constant<int, 1>{}); rd(std::integral_constant<int, 2>{});
        rd(std::integral_constant<int, 3>{}); rd(std::integral_constant<int, 4>{});
        static_assert(RD == 5, "the five reads above");
        auto step = [&](auto STc) __attribute__((always_inline)) {
            constexpr int st = decltype(STc)::value;
            constexpr int j = st / KH, kh = st - KH * j;
            if constexpr (st + RD < NSTEP) rd(std::integral_constant<int, st + RD>{});
            if constexpr (st >= 4 && (st - 4) % 20 == 0 && (st - 4) / 20 < G::PPW) dma_piece(std::integral_constant<int, (st - 4) / 20>{}, slot ^ 1);
            d3k_mma<DT, (kh < KHA), kh == 0>(cacc[j & 1][0], A[kh][0], bq[st % RING]);
            d3k_mma<DT, (kh < KHA), kh == 0>(cacc[j & 1][1], A[kh][1], bq[st % RING]);
            // (the fragment of two steps ago stays alive up to here: write-after-read on an MFMA source operand, d3c_conv.hpp)
            if constexpr (st >= 2) asm volatile("" ::"v"(bq[(st - 2) % RING]), "v"(cacc[j & 1][1]));
            if constexpr (j >= 1) {                                 // block j - 1 is finished under this block's MFMAs
                typedef std::integral_constant<int, j - 1> JP;
                // The compiler cannot see that the asm statements are MFMAs: nothing keeps it from scheduling the vector instructions that read
                // block j - 1's accumulators directly behind that block's last MFMA, inside the matrix pipe's latency (fp16 build: three
                // blocks of every tile came out wrong). This volatile statement is ordered against the per-step anchors above, so the reads
                // stay behind two steps (four MFMAs, 64 cycles) of this block.
                // The distance is ENFORCED, not assumed: only each step's [1]-MFMA is ordered against the volatile anchors, the
                // [0]-MFMAs are free to sink - hipcc 7.2 moved block j - 1's last MFMA to within 2 - 3 instructions of the first read. The
                // wait states of the matrix pipe (XDL write -> vector read: passes + 3 on gfx950, 11 for an 8-pass instruction) are
                // therefore part of this statement, which the accumulators' data dependence pins between the block's last MFMA and
                // the first read; tests/test_isa_hazards.py checks the emitted code. 12 cycles per 72-MFMA block.
                if constexpr (kh == 2) asm volatile(D3K_MFMA_PAD : "+v"(cacc[(j - 1) & 1][0]), "+v"(cacc[(j - 1) & 1][1]));
                if constexpr (kh == 2) epi_part(HRc, JP{}, std::integral_constant<int, 0>{}, guard);
                if constexpr (kh == 10) epi_part(HRc, JP{}, std::integral_constant<int, 1>{}, guard);
                if constexpr (kh == 18) epi_part(HRc, JP{}, std::integral_constant<int, 2>{}, guard);
                if constexpr (kh == 26) epi_part(HRc, JP{}, std::integral_constant<int, 3>{}, guard);
                if constexpr (kh == 30) epi_store(j - 1, ch0, mTile, mEnd);
            }
        };
        auto run = [&](auto... Sc) __attribute__((always_inline)) { (step(Sc), ...); };
        auto twelve = [&](auto Bc) __attribute__((always_inline)) {     // steps 12 B .. 12 B + 11
            constexpr int B0 = decltype(Bc)::value * 12;
            run(std::integral_constant<int, B0>{}, std::integral_constant<int, B0 + 1>{}, std::integral_constant<int, B0 + 2>{},
                std::integral_constant<int, B0 + 3>{}, std::integral_constant<int, B0 + 4>{}, std::integral_constant<int, B0 + 5>{},
                std::integral_constant<int, B0 + 6>{}, std::integral_constant<int, B0 + 7>{}, std::integral_constant<int, B0 + 8>{},
                std::integral_constant<int, B0 + 9>{}, std::integral_constant<int, B0 + 10>{}, std::integral_constant<int, B0 + 11>{});
        };
        auto block = [&](auto Jc) __attribute__((always_inline)) {      // the 36 steps of pixel block J
            constexpr int B0 = decltype(Jc)::value * 3;
            twelve(std::integral_constant<int, B0>{}); twelve(std::integral_constant<int, B0 + 1>{}); twelve(std::integral_constant<int, B0 + 2>{});
        };
        block(std::integral_constant<int, 0>{}); block(std::integral_constant<int, 1>{}); block(std::integral_constant<int, 2>{});
        block(std::integral_constant<int, 3>{}); block(std::integral_constant<int, 4>{}); block(std::integral_constant<int, 5>{});
        block(std::integral_constant<int, 6>{});
        {                                                           // the last block's epilogue: on its own
            // The compiler cannot see that the inline asm above are MFMAs: the wait states a vector instruction needs before it reads an
            // accumulator the matrix pipe is still writing are inserted by hand (the interleaved parts read theirs two steps behind the
            // block's last MFMA: far enough). Without them the last pixel block of every tile came out wrong.
            // (tied to the accumulators: a free-standing asm is not ordered against the non-volatile MFMA statements)
            asm volatile(D3K_MFMA_PAD : "+v"(cacc[(NB - 1) & 1][0]), "+v"(cacc[(NB - 1) & 1][1]));
            typedef std::integral_constant<int, NB - 1> JL;
            epi_part(HRc, JL{}, std::integral_constant<int, 0>{}, guard); epi_part(HRc, JL{}, std::integral_constant<int, 1>{}, guard);
            epi_part(HRc, JL{}, std::integral_constant<int, 2>{}, guard); epi_part(HRc, JL{}, std::integral_constant<int, 3>{}, guard);
            epi_store(NB - 1, ch0, mTile, mEnd);
        }
        guard.commit(p.ovf);
    };
    static_assert(4 + 20 * (G::PPW - 1) > KH * (NB - 2) + 30 && 4 + 20 * (G::PPW - 1) < KH * (NB - 1) + 30,
                  "behind the last piece (step 224) exactly two stores are issued: block 5's at step 246 and the last block's (the wait below)");

    // ---- prologue: the first tile's patch ----
    int chTile, n, y0;
    tile_pos(T.tile0, chTile, n, y0);
    dma_setup(true, n, y0);
    {
        auto all = [&](auto... Ic) __attribute__((always_inline)) { (dma_piece(Ic, 0), ...); };
        all(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{}, std::integral_constant<int, 3>{},
            std::integral_constant<int, 4>{}, std::integral_constant<int, 5>{}, std::integral_constant<int, 6>{}, std::integral_constant<int, 7>{},
            std::integral_constant<int, 8>{}, std::integral_constant<int, 9>{}, std::integral_constant<int, 10>{}, std::integral_constant<int, 11>{});
    }

    int slot = 0, t = T.tile0;
    // a RUN of tiles that share their channel tile: weights and BN constants are loaded in front of the run (d3c_conv.hpp)
    while (t < T.tend) {
        load_weights(chTile);
        __builtin_amdgcn_s_waitcnt(0x0070);                        // vmcnt(0) lgkmcnt(0), visible to the compiler's wait-count pass
        d3q_sync();
        bool same;
        do {
            const int tn = t + T.tstride;
            const bool more = tn < T.tend;
            int chN = chTile, nN = 0, y0N = 0;
            if (more) tile_pos(tn, chN, nN, y0N);
            dma_setup(more, nN, y0N);
            const uint32_t slotoff = (uint32_t)(slot * G::SLOT);
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int q = 0; q < 3; ++q) bs[j][q] = ba[j][q] + slotoff;
            {
                const int ch0 = chTile * G::BM + wave * 32 + 8 * fq;
                const int mTile = (n * p.H + y0) * G::W;             // first pixel of the tile (whole image rows: flat NHWC index)
                const int mEnd = (n * p.H + (y0 + G::ROWS < p.H ? y0 + G::ROWS : p.H)) * G::W;
                if (has_res) tile_steps(std::true_type{}, slot, ch0, mTile, mEnd);
                else tile_steps(std::false_type{}, slot, ch0, mTile, mEnd);
            }
            // the next patch has landed (behind its last piece, step 224, only the stores of the last two blocks - and, with a skip tensor, its
            // long consumed residual loads - were issued); every wave is done with this patch
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            d3q_sync();
            same = more && chN == chTile;
            chTile = chN; n = nN; y0 = y0N;
            slot ^= 1;
            t = tn;
        } while (same);
    }
}
#endif  // __HIP_DEVICE_COMPILE__

template <int DT>
__global__ __launch_bounds__(256, 1) void d3k_kernel(const D3Params p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    d3k_body<DT>(p, smem);
#endif  // __HIP_DEVICE_COMPILE__
}

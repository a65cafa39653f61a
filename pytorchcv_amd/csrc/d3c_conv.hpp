// d3c_conv.hpp - dense 3x3 / stride 1 / pad 1 convolution with 64 INPUT channels on 56-pixel-wide maps (ResNet stage 1: three
// layers of ResNet-50, four of ResNet-18), gfx950 MFMA: weights in REGISTERS, the input patch staged ONCE per tile with its zero
// padding physically in LDS, no barrier and no padded-tap select inside a tile.
//
// Replaces: nn.Conv2d(64 -> Cout, 3x3, stride 1, padding 1) + nn.BatchNorm2d(eval) + activation of ConvBlock.forward
//           (reference pytorchcv/models/common/conv.py:278-286) at ResBottleneck.conv2 / ResBlock.conv1, conv2 of the 56 x 56
//           stage (resnet.py:49,56,120-127), plus the residual add + ReLU of basic-block units (resnet.py:227-228) in the
//           epilogue. Same K order (filter row, filter column; one 64-channel slice), same MFMA sequence per accumulator, same
//           epilogue arithmetic as d3q_conv.hpp / igemm_conv.hpp: bit-identical results.
//
// Why (round 4). With 64 input and 64 output channels the general kernels are bound by the L2 -> LDS path, not by the MFMA:
// a 64 x 448 tile pulls 27 KB per K-step (the activation rows three times - once per filter row - plus the weights) for 896
// MFMA cycles, i.e. 30 B/clk per CU where ~16-20 arrive (profiles/experiments/r04_d3w_loop.md); 84-88 us per layer at batch 256
// against ~45 us of HBM time and ~30 us of MFMA time. Here:
//   * the 9 x 64 x 64 weights (73.7 KB) of a 64-channel tile sit in the REGISTERS of the four waves as MFMA A fragments
//     (32 channels x 576 = 144 registers per lane; one wave per SIMD, 512-register budget) - loaded once per block;
//   * a tile is 4 output rows of one image (224 pixels x 64 channels); its 6 input rows are staged ONCE by LDS-DMA (42 KB for
//     all nine taps: 4.3 B/clk) into an image with a 64-row pitch per image row whose rows 0 and 57 are zero: the filter taps are
//     plain row offsets (r x 64 + q), out-of-image rows arrive as zeros from the DMA's range check - no masks, no selects;
//   * two patch slots: the next tile's pieces are issued between the K-halves of the current one; ONE barrier per tile.
// LDS row u of an image row holds pixel u - 1; 16-byte chunk slot s of LDS row R holds K-chunk s ^ (R & 7) (source-side swizzle,
// as everywhere); the pitch of 64 keeps R & 7 independent of the image row, so a lane's fragment address is a per-(block, filter
// column) constant + an immediate for the filter row.
#pragma once
#include <type_traits>
#include "pcv_common.hpp"
#include "igemm_conv.hpp"     // Mma<DT>
#include "d3q_conv.hpp"       // D3Params, D3Tiles, d3q_tiles

struct D3CCfg {
    static constexpr int THREADS = 256;
    static constexpr int W = 56, ROWS = 4;                   // map width; output rows per tile
    static constexpr int BM = 64, BP = ROWS * W;             // 64 channels x 224 pixels
    static constexpr int NBLK = BP / 16;                     // 14 pixel blocks, 7 per wave
    static constexpr int PITCH = 64;                         // LDS rows per staged image row
    static constexpr int SROWS = ROWS + 2;                   // staged image rows
    static constexpr int SLOT = SROWS * PITCH * 128;         // 49 152 B
    static constexpr int DUMP = 2 * SLOT;                    // 1 KB: where the two pieces that do not exist (42, 43) land
    static constexpr int LDS = DUMP + 1024;
    static constexpr int NPIECE = SROWS * (W / 8);           // 42 one-KB pieces per tile
    static constexpr int PPW = (NPIECE + 3) / 4;             // pieces per wave: 11 (waves 2, 3: 10)
};

#if defined(__HIP_DEVICE_COMPILE__)
template <int DT>
__device__ __forceinline__ void d3c_body(const D3Params& p, char* smem) {
    typedef D3CCfg G;
    typedef typename Mma<DT>::frag frag;
    typedef __attribute__((address_space(3))) char lds_char;
    typedef const __attribute__((address_space(3))) frag* lds_fptr;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wc = wave >> 1, wp = wave & 1;                  // 32 channels x 112 pixels per wave
    const int fr = lane & 15, fq = lane >> 4;
    const int lrow = lane >> 3;
    const uint32_t lds0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(size_t)PCV_LDS(smem));
    const D3Tiles T = d3q_tiles(p);
    if (T.nMine == 0) return;

    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    const bool has_res = p.res != nullptr;
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res), 0, has_res ? p.res_bytes : 0u, 0x00020000);
    const int tilesPerImage = (p.H + G::ROWS - 1) / G::ROWS;

    // ---- the zero columns of both patch slots (LDS rows 0 and 57..63 of every staged image row; the DMA never writes them) ----
    for (int i = threadIdx.x; i < 2 * G::SROWS * 8 * 8; i += G::THREADS) {           // 16-byte chunks
        const int chunk = i & 7, prow = (i >> 3) & 7, ir = i >> 6;                      // ir: staged image row of slot 0 / 1
        const int row = prow == 0 ? 0 : G::W + prow;                                    // 0, 57 .. 63
        *reinterpret_cast<__attribute__((address_space(3))) u32x4*>((size_t)(lds0 + (uint32_t)((ir * G::PITCH + row) * 128 + chunk * 16))) =
            (u32x4){0u, 0u, 0u, 0u};
    }

    // ---- weights: A fragments of this wave's 32 channels, all 18 K-halves, in registers (reloaded when the channel tile changes) ----
    frag A[18][2];
    f32x4 es0, es1, eh0, eh1;
    auto load_weights = [&](int chTile) __attribute__((always_inline)) {
#pragma unroll
        for (int kh = 0; kh < 18; ++kh)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const uint32_t row = (uint32_t)(chTile * G::BM + wc * 32 + i * 16 + fr);
                const uint32_t off = (row * (uint32_t)p.Kpad + (uint32_t)((kh >> 1) * 64 + (fq + 4 * (kh & 1)) * 8)) * 2u;     // rows past the blob: zeros
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, off, 0, 0);
                A[kh][i] = __builtin_bit_cast(frag, v);
            }
        const int ch0 = chTile * G::BM + wc * 32 + 8 * fq;
        const int chl = ch0 < p.Cout ? ch0 : 0;                  // pad channels: any valid entry (never stored)
        es0 = *reinterpret_cast<const f32x4*>(p.scale + chl); es1 = *reinterpret_cast<const f32x4*>(p.scale + chl + 4);
        eh0 = *reinterpret_cast<const f32x4*>(p.shift + chl); eh1 = *reinterpret_cast<const f32x4*>(p.shift + chl + 4);
    };

    // ---- fragment addresses: block j, filter column q (row offset q, swizzle key (px + q) & 7); filter row r = + r * 8192 bytes ----
    uint32_t ba[7][3];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const int pix = wp * 112 + 16 * j + fr;                  // pixel inside the tile
        const int py = pix / G::W, px = pix - py * G::W;
#pragma unroll
        for (int q = 0; q < 3; ++q)
            ba[j][q] = lds0 + (uint32_t)((py * G::PITCH + px + q) * 128 + ((fq ^ ((px + q) & 7)) << 4));
    }

    // ---- DMA: piece idx = wave + 4 i (i = 0 .. 10; 42 pieces): staged image row ry = idx / 7, pixels 8 (idx % 7) .. + 7 ----
    // Source: a descriptor over the tile's IMAGE only (base = image start, num_records = one image), per-lane offset = this lane's
    // pixel and K-chunk + (y0 - 1 + ry) image rows: a row above or below the image is out of range and arrives as zeros - no
    // validity branch. 7 pieces of 1 KB are exactly one image row, so the offset is linear in idx: voff = tile part + i * 4096.
    // Destination: LDS rows ry * 64 + 1 + 8 k + lrow. (The first form - row / column / validity arithmetic per piece, ~30 scalar
    // instructions with two branches - cost a wave that is alone on its SIMD ~100 cycles per K-half.)
    const uint32_t imgBytes = (uint32_t)(p.H * G::W * 128);
    const uint32_t lanesrc = (uint32_t)(wave * 1024 + lrow * 128 + (((lane & 7) ^ ((1 + lrow) & 7)) << 4));
    uint32_t vt = 0x80000000u;                                    // the next tile's offset of piece 0 (2^31: no next tile - zeros into the free slot)
    __amdgpu_buffer_rsrc_t xr = xrsrc;
    auto dma_setup = [&](bool more, int n, int y0) __attribute__((always_inline)) {
        xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.x)) + (size_t)n * imgBytes, 0, imgBytes, 0x00020000);
        vt = more ? lanesrc + (uint32_t)((y0 - 1) * (G::W * 128)) : 0x80000000u;
    };
    auto dma_piece = [&](auto Ic, int slot) __attribute__((always_inline)) {
        constexpr int I = decltype(Ic)::value;
        constexpr int C0 = (4 * I) / 7, C1 = (4 * I) % 7;         // idx / 7 = C0 + (C1 + wave >= 7)
        const int ry = C0 + (wave >= 7 - C1 ? 1 : 0);
        const int idx = wave + 4 * I;
        const bool real = I < 10 || wave < G::NPIECE - 40;         // pieces 42, 43 (i = 10 of waves 2, 3) do not exist: zeros into the dump KB
        const uint32_t dst = lds0 + (real ? (uint32_t)(slot * G::SLOT + 128 + idx * 1024 + ry * 1024) : (uint32_t)G::DUMP);
        const uint32_t voff = real ? vt + (uint32_t)(I * 4096) : 0x80000000u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_char*)(size_t)dst, 16, voff, 0, 0, 0);
    };
    auto tile_pos = [&](int t, int& chTile, int& n, int& y0) __attribute__((always_inline)) {
        chTile = t % p.nChTiles;
        const int pt = t / p.nChTiles;
        n = pt / tilesPerImage;
        y0 = (pt - n * tilesPerImage) * G::ROWS;
    };

    // K loop order: PIXEL BLOCK outer, K-half inner. A block's two accumulators (its 16 pixels x this wave's 32 channels) run
    // through all 18 K-halves (36 MFMAs; a 16x16x32 chain on one accumulator issues at the full rate), then the block is
    // finished and stored WHILE the next block's MFMAs run - the vector ALU work of the epilogue (1 960 of a tile's 7 580 cycles
    // when it ran behind the K loop of all seven blocks) goes under the matrix pipe, with two accumulator pairs instead of seven.
    // One fragment read per step (block j, K-half kh), requested RD steps ahead into a ring of fragment values.
    constexpr int RD = 5, RING = 8, NSTEP = 7 * 18;
    f32x4 cacc[2][2];                                             // [block parity][channel half]
    frag bq[RING];
    uint32_t bs[7][3];                                            // ba + this tile's slot (set per tile)
    auto rd = [&](auto STc) __attribute__((always_inline)) {
        constexpr int st = decltype(STc)::value;
        constexpr int j = st / 18, kh = st - 18 * j, ks = kh >> 1, r = ks / 3, q = ks - 3 * r, h = kh & 1;
        const uint32_t a = bs[j][q] ^ (uint32_t)(h << 6);                           // (kc + 4) ^ key = (kc ^ key) ^ 4: bit 6 of the address
        bq[st % RING] = *reinterpret_cast<lds_fptr>((size_t)a + (size_t)(r * G::PITCH * 128));
    };
    // Activations none / ReLU / ReLU6 as BRANCH-FREE clamps to launch-uniform bounds (-inf / 0, +inf / 6; IEEE-754-2019 maximum /
    // minimum: a NaN stays a NaN, max(v, -inf) = v and min(v, +inf) = v bit for bit): the uniform switch of `clampn` is a dozen scalar
    // branches per pixel block, and a wave that is alone on its SIMD pays every one of them (3 100 cycles per tile's epilogue).
    const float alo = (p.act == PCV_ACT_RELU || p.act == PCV_ACT_RELU6) ? 0.f : -INFINITY, ahi = p.act == PCV_ACT_RELU6 ? 6.f : INFINITY;
    const float plo = (p.post_act == PCV_ACT_RELU || p.post_act == PCV_ACT_RELU6) ? 0.f : -INFINITY, phi = p.post_act == PCV_ACT_RELU6 ? 6.f : INFINITY;
    const float clo = alo > plo ? alo : plo, chi = ahi < phi ? ahi : phi;
    // One pixel block's epilogue, in FOUR parts of two values each (v = acc * scale + shift -> act -> (+ residual) -> post_act ->
    // packed pair; d3q_conv.hpp) + the 16-byte NHWC store. The parts of block j - 1 are written BETWEEN the steps of block j (its
    // accumulator pair is the other one), so that the vector ALU work sits under the next block's MFMAs in program order too - left
    // to the scheduler, a block's epilogue ran as one run of ~45 VALU instructions behind its last MFMA, the matrix pipe idle.
    u32x4 rrq[7];                                                 // the tile's residual pieces (requested at the tile's start)
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
            // nothing between the two activations: clamp(clamp(v, alo, ahi), plo, phi) = clamp(v, max(alo, plo), min(ahi, phi)) for
            // these bounds (0 / -inf below, 6 / +inf above), NaN and signed zero included
            v0 = __builtin_elementwise_minimum(__builtin_elementwise_maximum(v0, clo), chi);
            v1 = __builtin_elementwise_minimum(__builtin_elementwise_maximum(v1, clo), chi);
        }
        guard.see2(v0, v1);
        opend[e] = pack2<DT>(v0, v1);
    };
    auto epi_store = [&](int j, int ch0, int mTile, int mEnd) __attribute__((always_inline)) {
        const int m = mTile + wp * 112 + 16 * j + fr;
        const uint32_t boff = (ch0 < p.Cout && m < mEnd) ? (uint32_t)((m * p.Ypitch + ch0) * 2) : 0x80000000u;     // (the host keeps y below 2 GiB)
        __builtin_amdgcn_raw_buffer_store_b128(opend, yrsrc, boff, 0, 0);
    };
    // one tile: 126 steps; the 11 pieces of the NEXT tile's patch go out at steps 4, 15, 26, ... (into the other slot: every wave left
    // it before the barrier that ended the last tile)
    auto tile_steps = [&](auto HRc, int slot, int ch0, int mTile, int mEnd) __attribute__((always_inline)) {
        constexpr bool HR = decltype(HRc)::value;
        F16Guard<DT> guard;
        if constexpr (HR) {
            // the skip tensor's pieces of the whole tile, in FRONT of this tile's LDS-DMA pieces: vmcnt retires in order, and a residual
            // load issued behind a piece would make its consumer wait for that piece's trip to HBM
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const int m = mTile + wp * 112 + 16 * j + fr;
                const uint32_t roff = (ch0 < p.Cout && m < mEnd) ? (uint32_t)((m * p.Cout + ch0) * 2) : 0x80000000u;
                rrq[j] = __builtin_amdgcn_raw_buffer_load_b128(rrsrc, roff, 0, 0);
            }
        }
        rd(std::integral_constant<int, 0>{}); rd(std::integral_constant<int, 1>{}); rd(std::integral_constant<int, 2>{});
        rd(std::integral_constant<int, 3>{}); rd(std::integral_constant<int, 4>{});
        static_assert(RD == 5, "the five reads above");
        // (every index below is a template constant: a `#pragma unroll` loop of 126 large steps was only partly unrolled and its
        // run-time indices sent the weight fragments to scratch)
        auto step = [&](auto STc) __attribute__((always_inline)) {
            constexpr int st = decltype(STc)::value;
            constexpr int j = st / 18, kh = st - 18 * j;
            if constexpr (st + RD < NSTEP) rd(std::integral_constant<int, st + RD>{});
#if !defined(D3C_EXP) || D3C_EXP < 2
            if constexpr (st == 4) dma_piece(std::integral_constant<int, 0>{}, slot ^ 1);
            if constexpr (st == 15) dma_piece(std::integral_constant<int, 1>{}, slot ^ 1);
            if constexpr (st == 26) dma_piece(std::integral_constant<int, 2>{}, slot ^ 1);
            if constexpr (st == 37) dma_piece(std::integral_constant<int, 3>{}, slot ^ 1);
            if constexpr (st == 48) dma_piece(std::integral_constant<int, 4>{}, slot ^ 1);
            if constexpr (st == 59) dma_piece(std::integral_constant<int, 5>{}, slot ^ 1);
            if constexpr (st == 70) dma_piece(std::integral_constant<int, 6>{}, slot ^ 1);
            if constexpr (st == 81) dma_piece(std::integral_constant<int, 7>{}, slot ^ 1);
            if constexpr (st == 92) dma_piece(std::integral_constant<int, 8>{}, slot ^ 1);
            if constexpr (st == 103) dma_piece(std::integral_constant<int, 9>{}, slot ^ 1);
            if constexpr (st == 114) dma_piece(std::integral_constant<int, 10>{}, slot ^ 1);
#endif
            if constexpr (kh == 0) {
                cacc[j & 1][0] = Mma<DT>::run(A[0][0], bq[st % RING], (f32x4){0.f, 0.f, 0.f, 0.f});
                cacc[j & 1][1] = Mma<DT>::run(A[0][1], bq[st % RING], (f32x4){0.f, 0.f, 0.f, 0.f});
            } else {
                cacc[j & 1][0] = Mma<DT>::run(A[kh][0], bq[st % RING], cacc[j & 1][0]);
                cacc[j & 1][1] = Mma<DT>::run(A[kh][1], bq[st % RING], cacc[j & 1][1]);
            }
            // keep the fragment of two steps ago alive up to here: its register is not handed to a new read directly behind the MFMAs
            // that read it (write-after-read on an MFMA source operand stalls the LDS instruction until the matrix pipe has taken it)
            // (anchored behind this step's MFMAs through their accumulator: an asm with only the fragment as operand floats up to the
            // fragment's definition and keeps nothing alive)
            if constexpr (st >= 2) asm volatile("" ::"v"(bq[(st - 2) % RING]), "a"(cacc[j & 1][1]));
#if !defined(D3C_EXP) || D3C_EXP < 1
            if constexpr (j >= 1) {                                 // block j - 1 is finished under this block's MFMAs
                typedef std::integral_constant<int, j - 1> JP;
                if constexpr (kh == 1) epi_part(HRc, JP{}, std::integral_constant<int, 0>{}, guard);
                if constexpr (kh == 5) epi_part(HRc, JP{}, std::integral_constant<int, 1>{}, guard);
                if constexpr (kh == 9) epi_part(HRc, JP{}, std::integral_constant<int, 2>{}, guard);
                if constexpr (kh == 13) epi_part(HRc, JP{}, std::integral_constant<int, 3>{}, guard);
                if constexpr (kh == 15) epi_store(j - 1, ch0, mTile, mEnd);
            }
#endif
        };
        auto run = [&](auto... Sc) __attribute__((always_inline)) { (step(Sc), ...); };
        auto block = [&](auto Jc) __attribute__((always_inline)) {      // the 18 steps of pixel block J
            constexpr int B0 = decltype(Jc)::value * 18;
            run(std::integral_constant<int, B0>{}, std::integral_constant<int, B0 + 1>{}, std::integral_constant<int, B0 + 2>{},
                std::integral_constant<int, B0 + 3>{}, std::integral_constant<int, B0 + 4>{}, std::integral_constant<int, B0 + 5>{},
                std::integral_constant<int, B0 + 6>{}, std::integral_constant<int, B0 + 7>{}, std::integral_constant<int, B0 + 8>{},
                std::integral_constant<int, B0 + 9>{}, std::integral_constant<int, B0 + 10>{}, std::integral_constant<int, B0 + 11>{},
                std::integral_constant<int, B0 + 12>{}, std::integral_constant<int, B0 + 13>{}, std::integral_constant<int, B0 + 14>{},
                std::integral_constant<int, B0 + 15>{}, std::integral_constant<int, B0 + 16>{}, std::integral_constant<int, B0 + 17>{});
        };
        block(std::integral_constant<int, 0>{}); block(std::integral_constant<int, 1>{}); block(std::integral_constant<int, 2>{});
        block(std::integral_constant<int, 3>{}); block(std::integral_constant<int, 4>{}); block(std::integral_constant<int, 5>{});
        block(std::integral_constant<int, 6>{});
#if !defined(D3C_EXP) || D3C_EXP < 1
        {                                                           // the last block's epilogue: on its own
            typedef std::integral_constant<int, 6> JL;
            epi_part(HRc, JL{}, std::integral_constant<int, 0>{}, guard); epi_part(HRc, JL{}, std::integral_constant<int, 1>{}, guard);
            epi_part(HRc, JL{}, std::integral_constant<int, 2>{}, guard); epi_part(HRc, JL{}, std::integral_constant<int, 3>{}, guard);
            epi_store(6, ch0, mTile, mEnd);
        }
#endif
        guard.commit(p.ovf);
    };

    // ---- prologue: the first tile's patch ----
    int chTile, n, y0;
    tile_pos(T.tile0, chTile, n, y0);
    dma_setup(true, n, y0);
    dma_piece(std::integral_constant<int, 0>{}, 0); dma_piece(std::integral_constant<int, 1>{}, 0); dma_piece(std::integral_constant<int, 2>{}, 0);
    dma_piece(std::integral_constant<int, 3>{}, 0); dma_piece(std::integral_constant<int, 4>{}, 0); dma_piece(std::integral_constant<int, 5>{}, 0);
    dma_piece(std::integral_constant<int, 6>{}, 0); dma_piece(std::integral_constant<int, 7>{}, 0); dma_piece(std::integral_constant<int, 8>{}, 0);
    dma_piece(std::integral_constant<int, 9>{}, 0); dma_piece(std::integral_constant<int, 10>{}, 0);

    int slot = 0, t = T.tile0;
    // A RUN of tiles that share their channel tile (all of a block's tiles unless the grid is capped: tests): the weights and BN constants
    // are loaded in front of the run, so that the tile loop contains no register-destination load of them (a conditional reload inside
    // it made the compiler wait for the previous tile's stores at the first MFMA of every tile).
    while (t < T.tend) {
        load_weights(chTile);
        // The BUILTIN wait (vmcnt(0) lgkmcnt(0)): the compiler's wait-count pass sees that every load has returned.
        __builtin_amdgcn_s_waitcnt(0x0070);
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
            for (int j = 0; j < 7; ++j)
#pragma unroll
                for (int q = 0; q < 3; ++q) bs[j][q] = ba[j][q] + slotoff;
#ifdef D3C_CYCLES      // diagnostic build (tests/tools/d3c_cycles.py): shader-cycle stamps of this block's third tile
            const bool stamp__ = p.dbg != nullptr && t == T.tile0 + 2 * T.tstride;
            uint64_t c0__ = 0, c1__ = 0, c2__ = 0;
            if (stamp__) c0__ = __builtin_amdgcn_s_memtime();
#endif
            {
                const int ch0 = chTile * G::BM + wc * 32 + 8 * fq;
                const int mTile = (n * p.H + y0) * G::W;             // first pixel of the tile (whole image rows: flat NHWC index)
                const int mEnd = (n * p.H + (y0 + G::ROWS < p.H ? y0 + G::ROWS : p.H)) * G::W;
                if (has_res) tile_steps(std::true_type{}, slot, ch0, mTile, mEnd);
                else tile_steps(std::false_type{}, slot, ch0, mTile, mEnd);
            }
#ifdef D3C_CYCLES
            if (stamp__) c2__ = c1__ = __builtin_amdgcn_s_memtime();
#endif
            // the next patch has landed: behind its last piece (step 114) exactly two stores were issued - block 5's at step 6 x 18 + 15 =
            // 123 and the last block's behind the loop (with a skip tensor the residual loads went out in FRONT of the pieces) - so
            // vmcnt(2) waits for the piece and for neither store (vmcnt(1) also waited for block 5's HBM store, per tile);
            // every wave is done with this patch
            static_assert(114 > 5 * 18 + 15 && 114 < 6 * 18 + 15, "exactly two stores (blocks 5 and 6) are issued behind the last DMA piece");
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            d3q_sync();
#ifdef D3C_CYCLES
            if (stamp__ && lane == 0) {
                const uint64_t c3 = __builtin_amdgcn_s_memtime();
                uint32_t* d = p.dbg + (blockIdx.x * 4 + wave) * 4;
                d[0] = (uint32_t)(c1__ - c0__); d[1] = (uint32_t)(c2__ - c1__); d[2] = (uint32_t)(c3 - c2__); d[3] = 1u;
            }
#endif
            same = more && chN == chTile;
            chTile = chN; n = nN; y0 = y0N;
            slot ^= 1;
            t = tn;
        } while (same);
    }
}
#endif  // __HIP_DEVICE_COMPILE__

template <int DT>
__global__ __launch_bounds__(256, 1) void d3c_kernel(const D3Params p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    d3c_body<DT>(p, smem);
#endif  // __HIP_DEVICE_COMPILE__
}

// d3w_conv.hpp - dense 3x3 / stride 1 / pad 1 convolution on gfx950 MFMA, LARGE block tiles: eight self-loading waves,
// two per SIMD, up to 128 accumulator registers each (256 ch x 224 px, 128 x 416, 128 x 224, 64 x 448 block tiles).
//
// Replaces: nn.Conv2d(3x3, stride 1, padding 1) + nn.BatchNorm2d(eval) + activation of ConvBlock.forward
//           (reference pytorchcv/models/common/conv.py:278-286) at `conv3x3_block` call sites
//           (resnet.py:49,56,120-127 - ResBlock / ResBottleneck.conv2 - vgg.py, preresnet.py), plus the residual add +
//           ReLU of basic-block units (resnet.py:227-228) in the epilogue. Same arithmetic, same K order and same packed
//           blob as d3q_conv.hpp / igemm_conv.hpp: results are bit-identical to both.
//
// Why a second 3x3 kernel (round 4). d3q_kernel gives the LDS-DMA issue to four loader waves; three waves per SIMD leave 168
// registers per wave, i.e. 32 x 112 wave tiles and 256 x 112 / 128 x 224 block tiles, and its K loop ran at ~1 200 TFLOP/s
// against 1 650-1 800 for this loop shape (tests/tools/micro/selfload_loop.cpp, same DMA pattern, same LDS footprint):
//   * a 256 x 224 tile pulls 23 B/clk per CU through L2 -> LDS at full MFMA rate where 256 x 112 needs 42 and 128 x 224 29;
//   * 64 x 112 wave tiles read 0.39 fragments per MFMA (32 x 112: 0.64);
//   * a barrier interval holds 28 MFMAs per wave (448 matrix-pipe cycles) instead of 14, so the fixed cost of a hand-over
//     between the two wave groups is paid half as often per FLOP.
// The price is that every wave issues DMA pieces again (~100 cycles each): they sit in the wave's READ intervals, i.e.
// under the MFMAs of the SIMD's other wave.
//
// Structure (GEMM view, LDS images, swizzle, filter-row reuse of the activation tile, padded-tap selects: d3q_conv.hpp).
//   * 512 threads, one block per CU. Waves 0-3 (group 0) and 4-7 (group 1), one of each per SIMD, run the same program ONE
//     barrier interval apart: {fragment reads of a K-half + a share of the DMA pieces | 4 x 7 (2 x 13) MFMAs}. Four intervals per
//     K-step; group 1's program is rotated by one interval so that both groups execute the same barriers.
//   * Ring: weight tiles of K-steps s, s + 1, s + 2 (the pieces of s + 2 are issued during s), activation tiles of groups g and
//     g + 1 (issued during the K-steps q = 0 and q = 1 of g). Counted `s_waitcnt vmcnt(N)`, N = the pieces a wave issued during the
//     current K-step, in front of the barrier that ends it; raw `s_barrier`; never a drain inside a tile.
//   * RAW: a wave waits for ITS pieces of K-step s + 1 before the barrier that ends K-step s; the first read of them comes behind it.
//     WAR: every fragment read is retired (lgkmcnt(0)) before the barrier that ends its interval; the slot of K-step s - 1 is
//     re-filled from the first interval of K-step s on, behind group 1's last reads of it (the last interval of K-step s - 1).
//   * Epilogue of a tile: both groups in the SAME interval (group 0 in front of its first reads of the next tile, group 1 behind
//     its last MFMAs), with the first two K-steps of the next tile already in the ring.
#pragma once
#include <type_traits>
#include "pcv_common.hpp"
#include "igemm_conv.hpp"     // Mma<DT>, FastDiv
#include "d3q_conv.hpp"       // D3Params, D3Tiles, d3q_tiles, d3q_sync

// WC x WP: wave grid (channels x pixels), 8 waves. CBW / PBW: 16-row blocks per wave (channels / pixels).
// NSA_: slots of the weight ring (3: the tile of K-step s + 2 is issued during K-step s; 2: the tile of s + 1, due at the end of s - for
//     block tiles whose activation slots leave no room for a third weight tile).
// TRIM: pixel blocks cut from the END of the tile: the waves still run PBW blocks each (the barrier interval is set by the longest
//     wave), but the tile covers - stages, stores - 16 TRIM pixels less. 128 x 416 (TRIM 2 of 4 x 7 blocks): two activation slots of
//     418 rows leave room for the three-slot weight ring that 128 x 448 has no LDS for.
// KS: K-halves (32 elements each) per barrier interval - 1: four intervals per K-step (wave tiles of 24-28 accumulators),
//     2: two intervals per K-step, both halves' fragments read at once (wave tiles of 14 accumulators: 14 MFMAs are too short an interval).
template <int WC, int WP, int CBW, int PBW, int KS, int NSA_ = 3, int TRIM = 0> struct D3WCfg {
    static constexpr int THREADS = 512;
    static constexpr int BM = 16 * CBW * WC;                 // channel rows per block tile
    static constexpr int BP = 16 * (PBW * WP - TRIM);        // pixel rows per block tile (the last wave's last TRIM blocks lie outside it)
    static constexpr int NPA = BM / 8;                       // 1 KB DMA pieces (8 rows x 128 B) of one weight tile
    static constexpr int WLW = NPA / 8;                      // ... per wave
    static constexpr int BROWS = (BP + 2 + 7) / 8 * 8;       // rows of one activation tile: flat pixels P0 - 1 .. P0 + BP, padded
    static constexpr int NPB = BROWS / 8;
    static constexpr int XLW = (NPB + 7) / 8;                // activation pieces per wave per group
    static constexpr int NB0 = (XLW + 1) / 2, NB1 = XLW / 2; // ... issued during the group's K-steps q = 0 and q = 1
    static constexpr int ASZ = BM * 128;                     // bytes of one A slot
    static constexpr int BSZ = NPB * 1024;                   // bytes of one B slot
    static constexpr int NSA = NSA_, NSB = 2;                // weight ring: 3 slots = tiles issued two K-steps ahead, 2 slots = one
    static constexpr int ZOFF = (NSA * ASZ + NSB * BSZ + 2047) / 2048 * 2048;     // 2 KB of zeros, 2 KB-aligned (d3q_conv.hpp)
    static constexpr int DUMP = ZOFF + 2048;                 // 1 KB: where the (8 XLW - NPB) surplus pieces of a group land
    static constexpr int SSOFF = DUMP + 1024;                // fp32 BN scale [BM] | shift [BM] of the block's current channel tile
    static constexpr int LDS = SSOFF + 8 * BM;
    // weight pieces issued in the FIRST read interval of a K-step with NBQ activation pieces in it: the two intervals carry the same load
    static constexpr int wa0(int nbq) {
        if (KS == 2 || NSA == 2) return WLW;                 // one read interval per K-step / weights due at the end of THIS K-step: all of them first
        const int half = (WLW + nbq + 1) / 2 - nbq;
        return half < 0 ? 0 : (half > WLW ? WLW : half);
    }
    static_assert(WC * WP == 8, "eight waves");
    static_assert(KS == 1 || KS == 2, "one or two K-halves per interval");
    static_assert(NSA == 2 || NSA == 3, "weight ring of two or three tiles");
    static_assert(NPA % 8 == 0 && CBW % 2 == 0, "weight pieces split evenly over the waves; channel pairs per wave");
    static_assert(LDS <= 160 * 1024, "three weight tiles + two activation tiles must fit the LDS");
    static_assert(XLW <= 10, "row masks of the activation pieces are packed 3 bits each into one register");
    static_assert(CBW * PBW <= 32, "at most 128 accumulator registers");
};

#if defined(__HIP_DEVICE_COMPILE__)
// One wave's whole persistent loop. GRP 0: waves 0-3; GRP 1: waves 4-7, one barrier interval behind (separate straight-line
// instantiations: a per-interval `if (group)` makes the register allocator join both groups' states).
template <int DT, int WC, int WP, int CBW, int PBW, int KS, int NSA_, int TRIM, int GRP>
__device__ __forceinline__ void d3w_body(const D3Params& p, char* smem, const int wave) {
    typedef D3WCfg<WC, WP, CBW, PBW, KS, NSA_, TRIM> G;
    constexpr int BM = G::BM, BP = G::BP, WLW = G::WLW, XLW = G::XLW, NSA = G::NSA, AHEAD = NSA - 1;
    typedef typename Mma<DT>::frag frag;
    typedef __attribute__((address_space(3))) char lds_char;

    const int lane = threadIdx.x & 63;
    const int wc = wave / WP, wp = wave % WP;
    const int fr = lane & 15, fq = lane >> 4;
    const int lrow = lane >> 3;
    const int cs = (lane & 7) ^ lrow;                         // K-chunk this lane fetches (source-side swizzle)
    const uint32_t lds0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(size_t)PCV_LDS(smem));
    const D3Tiles T = d3q_tiles(p);
    if (T.nMine == 0) return;
    const int nk = p.nk;
    const int K_total = T.nMine * nk;
#ifdef D3W_CYCLES      // diagnostic build (tests/tools/d3w_cycles.py): s_memrealtime (100 MHz) / s_memtime stamps of every block's phases
    const uint64_t rt0__ = __builtin_amdgcn_s_memrealtime(), cy0__ = __builtin_amdgcn_s_memtime();
    uint64_t rt1__ = 0, rt2__ = 0, cy1__ = 0, cy2__ = 0;
#endif

    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);
    // activations: descriptor base one image row BELOW x (d3q_conv.hpp: the scalar offset of a group reaches the row above with r = 0)
    const uint32_t rowBytes = (uint32_t)(p.W * p.Cin * 2);
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.x)) - rowBytes, 0, p.x_bytes + 2u * rowBytes, 0x00020000);

    // ---- DMA side: this wave owns weight pieces 8 i + wave (rows 8 (8 i + wave) + lrow) and activation pieces 8 j + wave ----
    // The cursor runs two K-steps (weights) / one group (activations) ahead of the compute side: K-step s issues K-step s + 2.
    int la_tile = T.tile0, la_k = 0, la_slot = 0;
    uint32_t woff0 = 0;
    auto setup_a = [&](int t) __attribute__((always_inline)) {
        const int chTile = t % p.nChTiles;
        woff0 = (uint32_t)(((chTile * BM + 8 * wave + lrow) * p.Kpad + cs * 8) * 2);       // rows past the blob: out of range -> zeros
    };
    const uint32_t wstep = (uint32_t)(64 * p.Kpad * 2);        // 8 pieces x 8 rows further down the blob
    auto dma_a = [&](auto I0c, auto I1c) __attribute__((always_inline)) {                // pieces [I0, I1) of the weight tile
        constexpr int I0 = decltype(I0c)::value, I1 = decltype(I1c)::value;
#pragma unroll
        for (int i = I0; i < I1; ++i) {
            const uint32_t dst = lds0 + (uint32_t)(la_slot * G::ASZ + (8 * i + wave) * 1024);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_char*)(size_t)dst, 16, woff0 + i * wstep, la_k * 128, 0, 0);
        }
    };
    auto advance_a = [&]() __attribute__((always_inline)) {
        la_slot = la_slot + 1 == NSA ? 0 : la_slot + 1;
        if (++la_k == nk) {
            la_k = 0;
            la_tile += T.tstride;
            if (la_tile < T.tend) setup_a(la_tile);
        }
    };
    // LDS row u = 8 (8 j + wave) + lrow <-> flat pixel P0 + u - 1; group (r, c) reads that pixel shifted by (r - 1) image rows.
    // Piece j of a wave is 64 pixels behind piece j - 1: ONE offset register (pbv0) + a uniform stride; a piece whose row lies outside
    // [0, M) or the tile has no valid image row at all (mask bits 0) and reads zeros (offset 2^31).
    int lb_tile = T.tile0, lb_r = 0, lb_c = 0, lb_slot = 0;
    uint32_t pbv0 = 0u;            // byte offset of piece 0's pixel (+ this lane's chunk)
    uint32_t vmask = 0u;           // 3 bits per piece: image row ho + r - 1 exists, r = 0, 1, 2
    const uint32_t pstep = (uint32_t)(64 * p.Cin * 2);
    // One division per tile: the image-relative pixel index of consecutive pieces advances by 64 mod HW with one conditional
    // subtraction (a division per piece, inside a read interval, cost registers the K loop does not have).
    const uint32_t step_hw = 64u - fastdiv(64u, p.div_hw) * (uint32_t)p.HW;               // 64 mod HW
    auto table_rows = [&](int t) __attribute__((always_inline)) {
        const int tileP0 = (t / p.nChTiles) * BP;
        const int u0 = 8 * wave + lrow;                          // piece j of this wave: LDS rows 8 (8 j + wave) + lrow = u0 + 64 j
        const int m0 = tileP0 + u0 - 1;                          // >= -1
        // image-relative index of piece 1's pixel (never negative), stepped back once for piece 0
        const uint32_t m1 = (uint32_t)(m0 + 64);
        const uint32_t rem1 = m1 - fastdiv(m1, p.div_hw) * (uint32_t)p.HW;
        uint32_t rem = rem1 >= step_hw ? rem1 - step_hw : rem1 + (uint32_t)p.HW - step_hw;
        pbv0 = (uint32_t)((m0 * p.Cin + cs * 8) * 2);
        vmask = 0u;
#pragma unroll
        for (int j = 0; j < XLW; ++j) {
            const int u = u0 + 64 * j;
            const int m = m0 + 64 * j;
            const bool ok = u < BP + 2 && m >= 0 && m < p.M;
            const uint32_t vm = (rem >= (uint32_t)p.W ? 1u : 0u) | 2u | (rem + (uint32_t)p.W < (uint32_t)p.HW ? 4u : 0u);
            vmask |= (ok ? vm : 0u) << (3 * j);
            rem += step_hw;
            rem = rem >= (uint32_t)p.HW ? rem - (uint32_t)p.HW : rem;
        }
    };
    auto dma_b = [&](auto J0c, auto J1c) __attribute__((always_inline)) {                // pieces [J0, J1) of the group
        constexpr int J0 = decltype(J0c)::value, J1 = decltype(J1c)::value;
        const uint32_t soff = (uint32_t)(lb_r * p.W * p.Cin + lb_c * 64) * 2u;
#pragma unroll
        for (int j = J0; j < J1; ++j) {
            const uint32_t dst = lds0 + (uint32_t)(8 * j + wave < G::NPB ? NSA * G::ASZ + lb_slot * G::BSZ + (8 * j + wave) * 1024 : G::DUMP);
            const uint32_t t = (uint32_t)__builtin_amdgcn_sbfe((int)vmask, 3 * j + lb_r, 1);    // all ones: the image row exists
            const uint32_t voff = (t & (pbv0 + (uint32_t)j * pstep)) | (~t & 0x80000000u);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_char*)(size_t)dst, 16, voff, soff, 0, 0);
        }
    };
    auto advance_b = [&]() __attribute__((always_inline)) {   // group order inside a tile: (r, c), c fastest
        lb_slot ^= 1;
        if (++lb_c == p.slices) {
            lb_c = 0;
            if (++lb_r == 3) {
                lb_r = 0;
                lb_tile += T.tstride;
                if (lb_tile < T.tend) table_rows(lb_tile);
            }
        }
    };
    typedef std::integral_constant<int, 0> C0;
    typedef std::integral_constant<int, G::NB0> CB0;
    typedef std::integral_constant<int, XLW> CBN;
    typedef std::integral_constant<int, WLW> CAN;

    // ---- compute side ----
    f32x4 acc[CBW][PBW];
    frag a[KS][CBW], b[KS][PBW];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < CBW; ++i)
#pragma unroll
            for (int j = 0; j < PBW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    };
    const uint32_t afrag = lds0 + (uint32_t)((wc * 16 * CBW + fr) * 128);
    const uint32_t brow0 = (uint32_t)(wp * 16 * PBW + fr);
    uint32_t hm0 = 0, hm2 = 0;     // bit j: this lane's output pixel of block j is in image column 0 / W - 1 (per tile)
    const uint32_t step_w = 16u - fastdiv(16u, p.div_w) * (uint32_t)p.W;                  // 16 mod W
    auto set_masks = [&](int t) __attribute__((always_inline)) {                        // one division per tile (blocks are 16 pixels apart)
        const uint32_t m = (uint32_t)((t / p.nChTiles) * BP + wp * 16 * PBW + fr);
        uint32_t wo = m - fastdiv(m, p.div_w) * (uint32_t)p.W;                            // (n H + ho) W + wo = m
        hm0 = 0u;
        hm2 = 0u;
#pragma unroll
        for (int j = 0; j < PBW; ++j) {
            hm0 |= (wo == 0u ? 1u : 0u) << j;
            hm2 |= (wo + 1u == (uint32_t)p.W ? 1u : 0u) << j;
            wo += step_w;
            wo = wo >= (uint32_t)p.W ? wo - (uint32_t)p.W : wo;
        }
    };
    typedef const __attribute__((address_space(3))) char* lds_cptr;
    typedef const __attribute__((address_space(3))) frag* lds_fptr;
    // K-half h of the K-step in A slot sa / B slot sb, filter column Q, into fragment set u (fragment i / j sits i / j * 2048 bytes
    // behind the wave's first row)
    auto reads = [&](auto Uc, int sa, int sb, auto Qc, int h) __attribute__((always_inline)) {
        constexpr int Q = decltype(Qc)::value, U = decltype(Uc)::value;
#if defined(D3W_CYCLES) && defined(PCV_DBG_FLAGS)
        if (p.dbgflags & 128) return;                            // (bit 128: no fragment reads)
#endif
        const uint32_t abase = afrag + (uint32_t)(sa * G::ASZ);
        const uint32_t brow = brow0 + Q;
        const uint32_t bbase = lds0 + (uint32_t)(NSA * G::ASZ + sb * G::BSZ) + brow * 128u;
        const uint32_t zrow = lds0 + (uint32_t)G::ZOFF;
        const uint32_t kc = (uint32_t)(fq + 4 * h);
        lds_cptr ap = (lds_cptr)(size_t)(abase + ((kc ^ (uint32_t)(fr & 7)) << 4));
        lds_cptr bp = (lds_cptr)(size_t)(bbase + ((kc ^ (brow & 7u)) << 4));
        const uint32_t zsel = zrow + ((uint32_t)(size_t)bp & 2047u);        // the zero block through this lane's own banks
#pragma unroll
        for (int i = 0; i < CBW; ++i) a[U][i] = *reinterpret_cast<lds_fptr>(ap + i * 2048);
#pragma unroll
        for (int j = 0; j < PBW; ++j) {
            lds_cptr bj = bp;
            if constexpr (Q != 1) {
                const uint32_t t = (uint32_t)__builtin_amdgcn_sbfe((int)(Q == 0 ? hm0 : hm2), j, 1);   // all ones: horizontally padded tap
                bj = (lds_cptr)(size_t)((t & (zsel - (uint32_t)(j * 2048))) | (~t & (uint32_t)(size_t)bp));
            }
            b[U][j] = *reinterpret_cast<lds_fptr>(bj + j * 2048);
        }
    };
    auto mfmas = [&]() __attribute__((always_inline)) {
#if defined(D3W_CYCLES) && defined(PCV_DBG_FLAGS)
        if (p.dbgflags & 256) return;                            // (bit 256: no MFMAs)
#endif
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int u = 0; u < KS; ++u)
#pragma unroll
            for (int j = 0; j < PBW; ++j)
#pragma unroll
                for (int i = 0; i < CBW; ++i) acc[i][j] = Mma<DT>::run(a[u][i], b[u][j], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
    };
    auto reads_done = [&]() __attribute__((always_inline)) { __builtin_amdgcn_s_waitcnt(0xC07F); };      // lgkmcnt(0) only

    // BN scale / shift of the block's channel tile live in LDS ([BM] scale | [BM] shift, fp32): held in registers across the K loop they
    // cost 32 registers the 256-register budget does not have (scratch traffic in every interval); fetched from global memory inside
    // the epilogue they stood between two tiles' K loops.
    auto fill_ss = [&](int t) __attribute__((always_inline)) {
        const int chTile = t % p.nChTiles;
#pragma unroll
        for (int i = (int)threadIdx.x; i < 2 * BM; i += G::THREADS) {
            const int c = i < BM ? i : i - BM;
            const int ch = chTile * BM + c;
            const float v = (i < BM ? p.scale : p.shift)[ch < p.Cout ? ch : 0];          // pad channels: any valid entry (never stored)
            *reinterpret_cast<__attribute__((address_space(3))) float*>((size_t)(lds0 + G::SSOFF + 4 * i)) = v;
        }
    };
    // Epilogue (d3q_conv.hpp): v = acc * scale + shift -> act -> (+ residual) -> post_act -> one 16-byte NHWC store per (channel pair,
    // pixel block); branch-free, activations none / ReLU / ReLU6. Residual tiles are fetched four pixel blocks at a time.
    auto epilogue = [&](int t, auto HRc) __attribute__((always_inline)) {
        constexpr bool HR = decltype(HRc)::value;
        const ActClamp act = make_act(p.act), pact = make_act(p.post_act);
        const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res), 0, HR ? p.res_bytes : 0u, 0x00020000);
        const int chTile = t % p.nChTiles;
        const int tileP0 = (t / p.nChTiles) * BP;
        const int mBase = tileP0 + wp * 16 * PBW + fr;
        typedef const __attribute__((address_space(3))) f32x4* lds_f4;
        F16Guard<DT> guard;
#pragma unroll
        for (int ip = 0; ip < CBW / 2; ++ip) {
            const int cl = wc * 16 * CBW + 32 * ip + 8 * fq;    // channel inside the tile
            const int ch0 = chTile * BM + cl;
            const bool chok = ch0 < p.Cout;
            const f32x4 s0 = *(lds_f4)(size_t)(lds0 + G::SSOFF + 4 * cl), s1 = *(lds_f4)(size_t)(lds0 + G::SSOFF + 4 * cl + 16);
            const f32x4 h0 = *(lds_f4)(size_t)(lds0 + G::SSOFF + 4 * (BM + cl)), h1 = *(lds_f4)(size_t)(lds0 + G::SSOFF + 4 * (BM + cl) + 16);
#pragma unroll
            for (int j0 = 0; j0 < PBW; j0 += 4) {
                constexpr int JB = 4;
                u32x4 rr[JB];
                if constexpr (HR) {
#pragma unroll
                    for (int jj = 0; jj < JB; ++jj) {
                        const int j = j0 + jj;
                        if (j < PBW) {
                            const int m = mBase + 16 * j;
                            const uint32_t roff = (chok && m < p.M && (TRIM == 0 || m - tileP0 < BP)) ? (uint32_t)(((size_t)m * p.Cout + ch0) * 2) : 0x80000000u;
                            rr[jj] = __builtin_amdgcn_raw_buffer_load_b128(rrsrc, roff, 0, 0);
                        }
                    }
                }
#pragma unroll
                for (int jj = 0; jj < JB; ++jj) {
                    const int j = j0 + jj;
                    if (j < PBW) {
                        const int m = mBase + 16 * j;
                        float v[8];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = acc[2 * ip][j][e] * s0[e] + h0[e];
                            v[4 + e] = acc[2 * ip + 1][j][e] * s1[e] + h1[e];
                        }
                        clampn<8>(v, act);
                        if constexpr (HR) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float lo, hi;
                                unpack2<DT>(rr[jj][e], lo, hi);
                                v[2 * e] += lo;
                                v[2 * e + 1] += hi;
                            }
                        }
                        clampn<8>(v, pact);
                        if (TRIM == 0 || wp * PBW + j < PBW * WP - TRIM) guard.see(v);      // (a block past a trimmed tile's end holds garbage)
                        u32x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = pack2<DT>(v[2 * e], v[2 * e + 1]);
                        const bool ok = chok && m < p.M && (TRIM == 0 || m - tileP0 < BP);      // (trimmed tiles: the blocks past the tile's end)
                        const uint32_t boff = ok ? (uint32_t)(((size_t)m * p.Ypitch + ch0) * 2) : 0x80000000u;
                        __builtin_amdgcn_raw_buffer_store_b128(o, yrsrc, boff, 0, 0);
                    }
                }
            }
        }
        guard.commit(p.ovf);
    };

    // ---- prologue: weight tiles of K-steps 0 and 1, activation tile of group 0, the zero block, scale / shift ----
    setup_a(T.tile0);
    table_rows(T.tile0);
    if (wave < 2) *reinterpret_cast<__attribute__((address_space(3))) u32x4*>((size_t)(lds0 + G::ZOFF + (wave * 64 + lane) * 16)) = (u32x4){0u, 0u, 0u, 0u};
    dma_b(C0{}, CBN{});
    advance_b();
    dma_a(C0{}, CAN{});
    advance_a();
    if constexpr (AHEAD == 2) {
        dma_a(C0{}, CAN{});                                    // (nk >= 9: K-step 1 exists)
        advance_a();
    }
    fill_ss(T.tile0);
    zero_acc();
    set_masks(T.tile0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    d3q_sync();
#ifdef D3W_CYCLES
    rt1__ = __builtin_amdgcn_s_memrealtime(); cy1__ = __builtin_amdgcn_s_memtime();
#endif

    int sa = 0, sb = 0, k = 0, cur_tile = T.tile0;
    // One K-step (filter column Q of the current group), s = global K-step index of this block. KS = 1, four barrier intervals:
    //   group 0: reads K-half 0 + pieces | MFMAs | reads K-half 1 + pieces | MFMAs, wait
    //   group 1: MFMAs of K-step s - 1's second half (+ epilogue) | reads + pieces | MFMAs | reads + pieces, wait
    // KS = 2, two intervals: group 0: reads + pieces | MFMAs, wait;  group 1: MFMAs of K-step s - 1 (+ epilogue) | reads + pieces, wait.
    // Returns true after the tail (s == K_total: group 1's last MFMAs and both groups' last epilogue).
    auto kstep = [&](int s, auto Qc) __attribute__((always_inline)) -> bool {
        constexpr int Q = decltype(Qc)::value;
        constexpr int NBQ = Q == 0 ? G::NB0 : (Q == 1 ? G::NB1 : 0);
        constexpr int WA0 = G::wa0(NBQ);
        typedef std::integral_constant<int, WA0> CA0;
        typedef std::integral_constant<int, 0> U0;
        typedef std::integral_constant<int, KS - 1> U1;
        if constexpr (GRP == 1) {
            if (s > 0) mfmas();
        }
        bool refill = false;
        if constexpr (Q == 0) {                                // a tile ends behind q = 2 (nk is a multiple of 3)
            if (s > 0 && k == 0) {                             // the K-step before this one was its tile's last
                __builtin_amdgcn_sched_barrier(0);
#ifdef D3W_CYCLES
                if (s == K_total) { rt2__ = __builtin_amdgcn_s_memrealtime(); cy2__ = __builtin_amdgcn_s_memtime(); }
#endif
                if (p.res != nullptr) epilogue(cur_tile - T.tstride, std::true_type{});
                else epilogue(cur_tile - T.tstride, std::false_type{});
                zero_acc();
                set_masks(cur_tile);                            // (past the last tile: computed, never used)
                refill = p.nChTiles > 1 && s < K_total;
            }
            if (s == K_total) return true;
        }
#if defined(D3W_CYCLES) && defined(PCV_DBG_FLAGS)      // subtract-a-component timing (results wrong): dbg bit 32 = no activation pieces, 64 = no weight pieces in the loop
        const bool moreA = s + AHEAD < K_total && !(p.dbgflags & 64), moreB = s - Q + 3 < K_total && !(p.dbgflags & 32);
#else
        const bool moreA = s + AHEAD < K_total, moreB = s - Q + 3 < K_total;      // K-step s + AHEAD / group g + 1 exist
#endif
        auto issue_b = [&]() __attribute__((always_inline)) {
            if constexpr (Q == 0) { if (moreB) dma_b(C0{}, CB0{}); }
            if constexpr (Q == 1) { if (moreB) { dma_b(CB0{}, CBN{}); advance_b(); } }
        };
        // NSA = 3: activation pieces, then the first share of the weight pieces | the rest of the weight pieces.
        // NSA = 2: ALL weight pieces first (due at the end of this K-step) | the activation pieces (KS = 2: behind them in the one interval).
        auto issue0 = [&]() __attribute__((always_inline)) {
            if constexpr (AHEAD == 2) issue_b();
            if (moreA) dma_a(C0{}, CA0{});
            if constexpr (WA0 == WLW) { if (moreA) advance_a(); }
            if constexpr (AHEAD == 1 && KS == 2) issue_b();
        };
        auto issue1 = [&]() __attribute__((always_inline)) {
            if constexpr (WA0 < WLW) {
                if (moreA) {
                    dma_a(CA0{}, CAN{});
                    advance_a();
                }
            }
            if constexpr (AHEAD == 1 && KS == 1) issue_b();
        };
        // NSA = 3: everything issued BEFORE this K-step has landed. NSA = 2: also this K-step's weight pieces (the activation pieces issued
        // behind them stay in flight).
        auto wait_v = [&]() __attribute__((always_inline)) {
            if constexpr (AHEAD == 2) {
                if (moreA && (NBQ == 0 || moreB)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WLW + NBQ) : "memory");
                else if (moreA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WLW) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                if (NBQ != 0 && moreB) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBQ) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        };
        // the next tile's scale / shift, once every wave is past the epilogue that read the old ones (channel tiles differ between a
        // block's tiles only when the grid is not a multiple of the channel tile count: capped test grids)
        auto refill_ss = [&]() __attribute__((always_inline)) { if (refill) fill_ss(cur_tile); };
        if constexpr (GRP == 0 && KS == 1) {
            reads(U0{}, sa, sb, Qc, 0);
            issue0();
            reads_done();
            d3q_sync();
            refill_ss();
            mfmas();
            d3q_sync();
            reads(U0{}, sa, sb, Qc, 1);
            issue1();
            reads_done();
            d3q_sync();
            mfmas();
            wait_v();
            d3q_sync();
        } else if constexpr (GRP == 1 && KS == 1) {
            d3q_sync();
            refill_ss();
            reads(U0{}, sa, sb, Qc, 0);
            issue0();
            reads_done();
            d3q_sync();
            mfmas();
            d3q_sync();
            reads(U0{}, sa, sb, Qc, 1);
            issue1();
            reads_done();
            wait_v();
            d3q_sync();
        } else if constexpr (GRP == 0) {
            reads(U0{}, sa, sb, Qc, 0);
            reads(U1{}, sa, sb, Qc, 1);
            issue0();
            reads_done();
            d3q_sync();
            refill_ss();
            mfmas();
            wait_v();
            d3q_sync();
        } else {
            d3q_sync();
            refill_ss();
            reads(U0{}, sa, sb, Qc, 0);
            reads(U1{}, sa, sb, Qc, 1);
            issue0();
            reads_done();
            wait_v();
            d3q_sync();
        }
        if (++k == nk) {
            k = 0;
            cur_tile += T.tstride;
        }
        sa = sa + 1 == NSA ? 0 : sa + 1;
        if constexpr (Q == 2) sb ^= 1;
        return false;
    };
    for (int s = 0;; s += 3) {
        if (kstep(s, std::integral_constant<int, 0>{})) break;
        if (kstep(s + 1, std::integral_constant<int, 1>{})) break;
        if (kstep(s + 2, std::integral_constant<int, 2>{})) break;
    }
#ifdef D3W_CYCLES
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the last stores have left
    if (p.dbg != nullptr && lane == 0 && (wave == 0 || wave == 4)) {
        const uint64_t rt3 = __builtin_amdgcn_s_memrealtime(), cy3 = __builtin_amdgcn_s_memtime();
        uint32_t* d = p.dbg + (blockIdx.x * 2 + (wave >> 2)) * 8;
        d[0] = (uint32_t)rt0__; d[1] = (uint32_t)(rt1__ - rt0__); d[2] = (uint32_t)(rt2__ - rt0__); d[3] = (uint32_t)(rt3 - rt0__);
        d[4] = (uint32_t)(cy1__ - cy0__); d[5] = (uint32_t)(cy2__ - cy0__); d[6] = (uint32_t)(cy3 - cy0__); d[7] = (uint32_t)K_total;
    }
#endif
}
#endif  // __HIP_DEVICE_COMPILE__

template <int DT, int WC, int WP, int CBW, int PBW, int KS, int NSA_, int TRIM = 0>
__global__ __launch_bounds__(512, 2) void d3w_kernel(const D3Params p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // waves w and w + 4 share a SIMD (a workgroup's waves are dealt to the SIMDs cyclically)
    if (wave < 4) d3w_body<DT, WC, WP, CBW, PBW, KS, NSA_, TRIM, 0>(p, smem, wave);
    else d3w_body<DT, WC, WP, CBW, PBW, KS, NSA_, TRIM, 1>(p, smem, wave);
#endif  // __HIP_DEVICE_COMPILE__
}

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
template <int WC, int WP, int CBW, int PBW> struct D3WCfg {
    static constexpr int THREADS = 512;
    static constexpr int BM = 16 * CBW * WC;                 // channel rows per block tile
    static constexpr int BP = 16 * PBW * WP;                 // pixel rows per block tile
    static constexpr int NPA = BM / 8;                       // 1 KB DMA pieces (8 rows x 128 B) of one weight tile
    static constexpr int WLW = NPA / 8;                      // ... per wave
    static constexpr int BROWS = (BP + 2 + 7) / 8 * 8;       // rows of one activation tile: flat pixels P0 - 1 .. P0 + BP, padded
    static constexpr int NPB = BROWS / 8;
    static constexpr int XLW = (NPB + 7) / 8;                // activation pieces per wave per group
    static constexpr int NB0 = (XLW + 1) / 2, NB1 = XLW / 2; // ... issued during the group's K-steps q = 0 and q = 1
    static constexpr int ASZ = BM * 128;                     // bytes of one A slot
    static constexpr int BSZ = NPB * 1024;                   // bytes of one B slot
    static constexpr int NSA = 3, NSB = 2;
    static constexpr int ZOFF = (NSA * ASZ + NSB * BSZ + 2047) / 2048 * 2048;     // 2 KB of zeros, 2 KB-aligned (d3q_conv.hpp)
    static constexpr int DUMP = ZOFF + 2048;                 // 1 KB: where the (8 XLW - NPB) surplus pieces of a group land
    static constexpr int LDS = DUMP + 1024;
    // weight pieces issued in the FIRST read interval of a K-step with NBQ activation pieces in it: the two intervals carry the same load
    static constexpr int wa0(int nbq) {
        const int half = (WLW + nbq + 1) / 2 - nbq;
        return half < 0 ? 0 : (half > WLW ? WLW : half);
    }
    static_assert(WC * WP == 8, "eight waves");
    static_assert(NPA % 8 == 0 && CBW % 2 == 0, "weight pieces split evenly over the waves; channel pairs per wave");
    static_assert(LDS <= 160 * 1024, "three weight tiles + two activation tiles must fit the LDS");
    static_assert(XLW <= 10, "row masks of the activation pieces are packed 3 bits each into one register");
    static_assert(CBW * PBW <= 32, "at most 128 accumulator registers");
};

#if defined(__HIP_DEVICE_COMPILE__)
// One wave's whole persistent loop. GRP 0: waves 0-3; GRP 1: waves 4-7, one barrier interval behind (separate straight-line
// instantiations: a per-interval `if (group)` makes the register allocator join both groups' states).
template <int DT, int WC, int WP, int CBW, int PBW, int GRP>
__device__ __forceinline__ void d3w_body(const D3Params& p, char* smem, const int wave) {
    typedef D3WCfg<WC, WP, CBW, PBW> G;
    constexpr int BM = G::BM, BP = G::BP, WLW = G::WLW, XLW = G::XLW, NSA = G::NSA;
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
    const int K_total = T.nMine * nk, G_total = T.nMine * (nk / 3);

    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);
    // activations: descriptor base one image row BELOW x (d3q_conv.hpp: the scalar offset of a group reaches the row above with r = 0)
    const uint32_t rowBytes = (uint32_t)(p.W * p.Cin * 2);
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.x)) - rowBytes, 0, p.x_bytes + 2u * rowBytes, 0x00020000);

    // ---- DMA side: this wave owns weight pieces 8 i + wave (rows 8 (8 i + wave) + lrow) and activation pieces 8 j + wave ----
    int la_tile = T.tile0, la_k = 0, la_slot = 0, la_g = 0;    // la_g: global index of the next K-step to issue
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
        ++la_g;
        la_slot = la_slot + 1 == NSA ? 0 : la_slot + 1;
        if (++la_k == nk) {
            la_k = 0;
            la_tile += T.tstride;
            if (la_tile < T.tend) setup_a(la_tile);
        }
    };
    // LDS row u = 8 (8 j + wave) + lrow <-> flat pixel P0 + u - 1; group (r, c) reads that pixel shifted by (r - 1) image rows
    int lb_tile = T.tile0, lb_r = 0, lb_c = 0, lb_slot = 0, lb_g = 0;
    uint32_t pbv[XLW];             // byte offset of the pixel itself (+ this lane's chunk), or 2^31 for a row outside [0, M) / the tile
    uint32_t vmask = 0u;           // 3 bits per piece: image row ho + r - 1 exists, r = 0, 1, 2
    // One division per tile: consecutive pieces of a wave are 64 pixels apart, so the image-relative pixel index advances by
    // 64 mod HW with one conditional subtraction (the straightforward form - a division per piece, inside a read interval - cost
    // registers the K loop does not have).
    const uint32_t step_hw = 64u - fastdiv(64u, p.div_hw) * (uint32_t)p.HW;               // 64 mod HW
    auto table_rows = [&](int t) __attribute__((always_inline)) {
        const int tileP0 = (t / p.nChTiles) * BP;
        const int u0 = 8 * wave + lrow;                          // piece j of this wave: LDS rows 8 (8 j + wave) + lrow = u0 + 64 j
        const int m0 = tileP0 + u0 - 1;                          // >= -1
        // image-relative index of piece 1's pixel (never negative), stepped back once for piece 0
        const uint32_t m1 = (uint32_t)(m0 + 64);
        const uint32_t rem1 = m1 - fastdiv(m1, p.div_hw) * (uint32_t)p.HW;
        uint32_t rem = rem1 >= step_hw ? rem1 - step_hw : rem1 + (uint32_t)p.HW - step_hw;
        vmask = 0u;
#pragma unroll
        for (int j = 0; j < XLW; ++j) {
            const int u = u0 + 64 * j;
            const int m = m0 + 64 * j;
            const bool ok = u < BP + 2 && m >= 0 && m < p.M;
            const uint32_t vm = (rem >= (uint32_t)p.W ? 1u : 0u) | 2u | (rem + (uint32_t)p.W < (uint32_t)p.HW ? 4u : 0u);
            pbv[j] = ok ? (uint32_t)((m * p.Cin + cs * 8) * 2) : 0x80000000u;
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
            const uint32_t voff = (t & pbv[j]) | (~t & 0x80000000u);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_char*)(size_t)dst, 16, voff, soff, 0, 0);
        }
    };
    auto advance_b = [&]() __attribute__((always_inline)) {   // group order inside a tile: (r, c), c fastest
        ++lb_g;
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
    frag a[CBW], b[PBW];
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
    // K-half h of the K-step in A slot sa / B slot sb, filter column Q (fragment i / j sits i / j * 2048 bytes behind the wave's first row)
    auto reads = [&](int sa, int sb, auto Qc, int h) __attribute__((always_inline)) {
        constexpr int Q = decltype(Qc)::value;
        const uint32_t abase = afrag + (uint32_t)(sa * G::ASZ);
        const uint32_t brow = brow0 + Q;
        const uint32_t bbase = lds0 + (uint32_t)(NSA * G::ASZ + sb * G::BSZ) + brow * 128u;
        const uint32_t zrow = lds0 + (uint32_t)G::ZOFF;
        const uint32_t kc = (uint32_t)(fq + 4 * h);
        lds_cptr ap = (lds_cptr)(size_t)(abase + ((kc ^ (uint32_t)(fr & 7)) << 4));
        lds_cptr bp = (lds_cptr)(size_t)(bbase + ((kc ^ (brow & 7u)) << 4));
        const uint32_t zsel = zrow + ((uint32_t)(size_t)bp & 2047u);        // the zero block through this lane's own banks
#pragma unroll
        for (int i = 0; i < CBW; ++i) a[i] = *reinterpret_cast<lds_fptr>(ap + i * 2048);
#pragma unroll
        for (int j = 0; j < PBW; ++j) {
            lds_cptr bj = bp;
            if constexpr (Q != 1) {
                const uint32_t t = (uint32_t)__builtin_amdgcn_sbfe((int)(Q == 0 ? hm0 : hm2), j, 1);   // all ones: horizontally padded tap
                bj = (lds_cptr)(size_t)((t & (zsel - (uint32_t)(j * 2048))) | (~t & (uint32_t)(size_t)bp));
            }
            b[j] = *reinterpret_cast<lds_fptr>(bj + j * 2048);
        }
    };
    auto mfmas = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < PBW; ++j)
#pragma unroll
            for (int i = 0; i < CBW; ++i) acc[i][j] = Mma<DT>::run(a[i], b[j], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
    };
    auto reads_done = [&]() __attribute__((always_inline)) { __builtin_amdgcn_s_waitcnt(0xC07F); };      // lgkmcnt(0) only

    // Epilogue (d3q_conv.hpp): v = acc * scale + shift -> act -> (+ residual) -> post_act -> one 16-byte NHWC store per (channel pair,
    // pixel block); branch-free, activations none / ReLU / ReLU6.
    // The BN scale / shift of the wave's channels are loaded at the START of the epilogue (the fragment registers are dead there): held
    // across the K loop they cost 32 registers the 256-register budget does not have (scratch traffic in every interval).
    auto epilogue = [&](int t) __attribute__((always_inline)) {
        const ActClamp act = make_act(p.act), pact = make_act(p.post_act);
        const bool has_res = p.res != nullptr;
        const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res), 0, has_res ? p.res_bytes : 0u, 0x00020000);
        const int chTile = t % p.nChTiles;
        const int tileP0 = (t / p.nChTiles) * BP;
        const int mBase = tileP0 + wp * 16 * PBW + fr;
        f32x4 es0[CBW / 2], es1[CBW / 2], eh0[CBW / 2], eh1[CBW / 2];
#pragma unroll
        for (int ip = 0; ip < CBW / 2; ++ip) {
            const int ch0 = chTile * BM + wc * 16 * CBW + 32 * ip + 8 * fq;
            const int chl = ch0 < p.Cout ? ch0 : 0;              // table index of a pad channel: any valid one (never stored)
            es0[ip] = *reinterpret_cast<const f32x4*>(p.scale + chl); es1[ip] = *reinterpret_cast<const f32x4*>(p.scale + chl + 4);
            eh0[ip] = *reinterpret_cast<const f32x4*>(p.shift + chl); eh1[ip] = *reinterpret_cast<const f32x4*>(p.shift + chl + 4);
        }
        F16Guard<DT> guard;
#pragma unroll
        for (int ip = 0; ip < CBW / 2; ++ip) {
            const int ch0 = chTile * BM + wc * 16 * CBW + 32 * ip + 8 * fq;
            const bool chok = ch0 < p.Cout;
            const f32x4 s0 = es0[ip], s1 = es1[ip], h0 = eh0[ip], h1 = eh1[ip];
            u32x4 rr[PBW];
#pragma unroll
            for (int j = 0; j < PBW; ++j) rr[j] = (u32x4){0u, 0u, 0u, 0u};
            if (has_res) {
#pragma unroll
                for (int j = 0; j < PBW; ++j) {
                    const int m = mBase + 16 * j;
                    const uint32_t roff = (chok && m < p.M) ? (uint32_t)(((size_t)m * p.Cout + ch0) * 2) : 0x80000000u;
                    rr[j] = __builtin_amdgcn_raw_buffer_load_b128(rrsrc, roff, 0, 0);
                }
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
                guard.see(v);
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = pack2<DT>(v[2 * e], v[2 * e + 1]);
                const bool ok = chok && m < p.M;
                const uint32_t boff = ok ? (uint32_t)(((size_t)m * p.Ypitch + ch0) * 2) : 0x80000000u;
                __builtin_amdgcn_raw_buffer_store_b128(o, yrsrc, boff, 0, 0);
            }
        }
        guard.commit(p.ovf);
    };

    // ---- prologue: weight tiles of K-steps 0 and 1, activation tile of group 0, the zero block ----
    setup_a(T.tile0);
    table_rows(T.tile0);
    if (wave < 2) *reinterpret_cast<__attribute__((address_space(3))) u32x4*>((size_t)(lds0 + G::ZOFF + (wave * 64 + lane) * 16)) = (u32x4){0u, 0u, 0u, 0u};
    dma_b(C0{}, CBN{});
    advance_b();
    dma_a(C0{}, CAN{});
    advance_a();
    if (K_total > 1) {
        dma_a(C0{}, CAN{});
        advance_a();
    }
    zero_acc();
    set_masks(T.tile0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    d3q_sync();

    int sa = 0, sb = 0, k = 0, cur_tile = T.tile0, ep_tile = T.tile0;
    bool ep = false;
    // One K-step (filter column Q of the current group), s = global K-step index of this block. Four barrier intervals:
    //   group 0: reads K-half 0 + pieces | MFMAs | reads K-half 1 + pieces | MFMAs, wait
    //   group 1: MFMAs of K-step s - 1's second half (+ epilogue) | reads + pieces | MFMAs | reads + pieces, wait
    // Returns true after the tail (s == K_total: group 1's last MFMAs and both groups' last epilogue).
    auto kstep = [&](int s, auto Qc) __attribute__((always_inline)) -> bool {
        constexpr int Q = decltype(Qc)::value;
        constexpr int NBQ = Q == 0 ? G::NB0 : (Q == 1 ? G::NB1 : 0);
        constexpr int WA0 = G::wa0(NBQ);
        typedef std::integral_constant<int, WA0> CA0;
        if constexpr (GRP == 1) {
            if (s > 0) mfmas();
        }
        if constexpr (Q == 0) {                                // a tile ends behind q = 2 (nk is a multiple of 3)
            if (ep) {
                epilogue(ep_tile);
                zero_acc();
                set_masks(cur_tile);                            // (past the last tile: computed, never used)
            }
            if (s == K_total) return true;
        }
        const bool moreA = la_g < K_total, moreB = lb_g < G_total;      // K-step s + 2 / group g + 1 exist
        auto issue0 = [&]() __attribute__((always_inline)) {
            if constexpr (Q == 0) { if (moreB) dma_b(C0{}, CB0{}); }
            if constexpr (Q == 1) { if (moreB) { dma_b(CB0{}, CBN{}); advance_b(); } }
            if (moreA) dma_a(C0{}, CA0{});
        };
        auto issue1 = [&]() __attribute__((always_inline)) {
            if (moreA) {
                dma_a(CA0{}, CAN{});
                advance_a();
            }
        };
        auto wait_v = [&]() __attribute__((always_inline)) {   // everything issued BEFORE this K-step has landed
            if (moreA && (NBQ == 0 || moreB)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WLW + NBQ) : "memory");
            else if (moreA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WLW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
        if constexpr (GRP == 0) {
            reads(sa, sb, Qc, 0);
            issue0();
            reads_done();
            d3q_sync();
            mfmas();
            d3q_sync();
            reads(sa, sb, Qc, 1);
            issue1();
            reads_done();
            d3q_sync();
            mfmas();
            wait_v();
            d3q_sync();
        } else {
            d3q_sync();
            reads(sa, sb, Qc, 0);
            issue0();
            reads_done();
            d3q_sync();
            mfmas();
            d3q_sync();
            reads(sa, sb, Qc, 1);
            issue1();
            reads_done();
            wait_v();
            d3q_sync();
        }
        ep = (k == nk - 1);
        ep_tile = cur_tile;
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
}
#endif  // __HIP_DEVICE_COMPILE__

template <int DT, int WC, int WP, int CBW, int PBW>
__global__ __launch_bounds__(512, 2) void d3w_kernel(const D3Params p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // waves w and w + 4 share a SIMD (a workgroup's waves are dealt to the SIMDs cyclically)
    if (wave < 4) d3w_body<DT, WC, WP, CBW, PBW, 0>(p, smem, wave);
    else d3w_body<DT, WC, WP, CBW, PBW, 1>(p, smem, wave);
#endif  // __HIP_DEVICE_COMPILE__
}

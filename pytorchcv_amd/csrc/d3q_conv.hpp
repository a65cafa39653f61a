// d3q_conv.hpp - dense 3x3 / stride 1 / pad 1 convolution on gfx950 MFMA with filter-ROW reuse of the activation operand.
// The MFMA-bound class of the path (ResNet-50: 16 layers, 48 % of the MACs).
//
// Replaces: nn.Conv2d(3x3, stride 1, padding 1) + nn.BatchNorm2d(eval) + activation of ConvBlock.forward
//           (reference pytorchcv/models/common/conv.py:278-286) at `conv3x3_block` call sites
//           (resnet.py:49,56,120-127 - ResBlock / ResBottleneck.conv2 - vgg.py, preresnet.py), plus the residual add +
//           ReLU of basic-block units (resnet.py:227-228) in the epilogue.
//
// GEMM view: Y^T[ch, pixel] = sum_k Wp[ch, k] X[pixel, k], k = (filter row r, 64-channel slice c, filter column q) - the
// packed blob of `plan_conv`'s `conv3` order. A = weights (one 128-byte row per channel and K-step), B = pixels.
//
// What the stamps of the first 8-wave kernel (d3x3, im2col gather per K-step) showed: a K-step's LDS-DMA (58 KB for a
// 256 x 208 tile) passes the texture path at ~1 KB per 16 clk per CU and the issuing waves block on it - DMA time and MFMA
// time ADD UP (3 400 cycles per K-step for 1 664 of MFMA), and a 2-deep ring leaves no interval in which pieces can be
// issued without standing in front of somebody's MFMAs. Two thirds of those bytes are the same input rows fetched again for
// the next filter column. So here:
//   * B is staged ONCE per group g = (r, c): the BP + 2 input rows of the flat pixel range [P0 - 1, P0 + BP] shifted by
//     (r - 1) image rows (vertical padding = per-row out-of-range DMA offsets, as before). The three K-steps q = 0, 1, 2 of
//     the group read it at row offsets 0, 1, 2. Horizontal padding - the left neighbour of a pixel in column 0 is the
//     previous image row's last pixel in this flat range - is applied at fragment-read time: a lane whose output pixel sits
//     in column 0 (q = 0) or W - 1 (q = 2) reads a zero row instead (address select, two VALU per fragment; q = 1 none).
//     DMA bytes per K-step: 256 x 208 tile 58 -> 41 KB, 128 x 416 68 -> 34 KB, 64 x 448 64 -> 27 KB.
//   * The LDS this frees holds a 3-deep ring of weight tiles (A): the pieces of K-step k + 2 and of group g + 1 are issued
//     during K-step k, two K-steps ahead of their first use; `s_waitcnt vmcnt(N)` with N = the pieces issued during the
//     current K-step, raw `s_barrier`, never a drain.
//   * 768 threads, one block per CU: EIGHT COMPUTE waves + FOUR LOADER waves (three waves per SIMD, 168 registers each).
//     Stamps of the self-loading variants put one LDS-DMA piece at ~100 cycles of its issuing wave whatever the placement
//     (in front of the MFMAs, beside the fragment reads, behind them): 6 pieces per wave and K-step stretched every interval
//     that carried them from ~520 to ~1 020 cycles. A loader wave pays that price in cycles nobody else needs - its VMEM issue
//     runs beside the compute waves' MFMA and LDS issue - and the compute waves' loop has no VMEM instruction left in it.
//     Compute waves 0-3 and 4-7 (one of each per SIMD) run one barrier interval apart - while one group reads fragments, the
//     other's MFMAs own the matrix pipe (MI355X_MICROARCH.md, "Two waves per SIMD"). KS K-halves per section: a K-step is
//     4 (KS = 1) or 2 (KS = 2) barrier intervals; the loaders take part in every barrier.
//   * Tile shapes chosen by the host so that the tile count fills whole rounds of the CUs (wave tiles of 7 or 13 pixel
//     blocks: 112 / 208 / 416 / 448-pixel tiles).
//
// Barrier/visibility rules (cdna_hip_programming.md, "Read a staged buffer one phase AFTER the wait that retires it"):
//   RAW  a loader waits for ITS pieces of K-step k + 1 / group g + 1 (counted vmcnt) before the barrier that ends K-step k;
//        their first read is issued after that barrier.
//   WAR  every fragment read is retired (lgkmcnt(0)) before the barrier that ends its interval; the A slot of K-step k - 1
//        and the B slot of group g - 1 are re-filled from K-step k / 3 g on, after both groups' last reads of them.
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
    int nk;                 // K-steps = 9 * Cin / 64 (a multiple of 3)
    int slices;             // Cin / 64
    int Kpad;
    int act, post_act;
    int nChTiles, nTiles;
    uint32_t* dbg;          // diagnostic builds only (-DD3X3_STAMPS)
    int stride, Hin, Win;   // 1x1 mode only: output pixel (n, ho, wo) reads input pixel (n, stride ho, stride wo) of an Hin x Win map
                            // (H, W, HW, div_hw, div_w then describe the OUTPUT map)
    uint32_t* ovf;          // the context's fp16 overflow counter (pcv_common.hpp, F16Guard)
    int tailN;              // p1r_conv.hpp only: tiles nTiles .. nTiles + tailN - 1 are split into 16-pixel units over the blocks (0: none)
    int dbgflags;           // timing experiments only (pcv_set_tuning("dbg", bits); results are WRONG with any bit set): 1 = output stores
                            // dropped (out-of-range offsets), 2 = no epilogue at all, 4 = loaders keep the first tile's row table, 8 = compute waves keep the
                            // first tile's column masks, 16 = row table built at the tile change (A/B of the look-ahead build; results stay right)
};

// In-kernel stamps (cdna_hip_programming.md section 7): a diagnostic build (-DD3X3_STAMPS) times ONE section per K-step - the
// stamp that opens it and the stamp that closes it, s_memtime low words written into the lanes of one VGPR per wave - and
// rotates the section from K-step to K-step (steps 4..30: every section three times). Stamp points sit behind a barrier or an
// explicit lgkmcnt(0), where no LDS read is outstanding (s_memtime returns through lgkmcnt). The product build compiles none
// of this.
// Timing experiments that produce WRONG results (pcv_set_tuning("dbg", bits)) exist only in diagnostic builds (make EXTRA=-DPCV_DBG_FLAGS);
// the product library compiles none of them and refuses the key.
#ifdef PCV_DBG_FLAGS
#define D3_DBG(bits) ((p.dbgflags & (bits)) != 0)
#else
#define D3_DBG(bits) false
#endif
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

// WC x WP: compute-wave grid (channels x pixels), 8 waves. CBW / PBW: 16-row blocks per wave (channels / pixels).
// KS: K-halves (32 elements each) per read / MFMA section.
// ONE: the 1x1 / stride 1 mode for K-heavy layers (ResNeXt-101 stage 3: 512 <-> 1024 channels at 14x14): every K-step is its own
// group - the activation tile of K-step s is the same pixel rows, 64-channel slice s - so the B ring is three slots deep like the A
// ring (tiles of K-step s + 2 are issued during K-step s) and the compute waves always take the unmasked centre-column path.
template <int WC, int WP, int CBW, int PBW, int KS, bool ONE = false> struct D3Cfg {
    static constexpr int NLOAD = 4;                          // loader waves (waves 8..11)
    static constexpr int THREADS = 64 * (8 + NLOAD);
    static constexpr int BM = 16 * CBW * WC;                 // channel rows per block tile
    static constexpr int BP = 16 * PBW * WP;                 // pixel rows per block tile
    static constexpr int NPA = BM / 8;                       // 1 KB DMA pieces (8 rows x 128 B) of one weight tile
    static constexpr int WLW = NPA / NLOAD;                  // ... per loader
    static constexpr int BROWS = (BP + 2 + 7) / 8 * 8;       // rows of one activation tile: flat pixels P0 - 1 .. P0 + BP, padded
    static constexpr int NPB = BROWS / 8;
    static constexpr int XLW = (NPB + NLOAD - 1) / NLOAD;    // activation pieces per loader per group
    static constexpr int NB0 = (XLW + 1) / 2, NB1 = XLW / 2; // ... issued during the group's K-steps q = 0 and q = 1
    static constexpr int ASZ = BM * 128;                     // bytes of one A slot
    static constexpr int BSZ = NPB * 1024;                   // bytes of one B slot
    static constexpr int NSA = 3;
    static constexpr int NSB = ONE ? 3 : 2;
    // 2 KB of zeros (16 rows), 2 KB-aligned: a horizontally padded tap reads the zero block at the SAME offset inside its 2 KB
    // window as the row it replaces, i.e. through the same LDS banks - one shared 16-byte zero chunk put 1.9 conflict cycles on
    // every fragment read of the 3x3 mode (SQ_LDS_BANK_CONFLICT: 4.5 M per launch; the 1x1 mode, which never masks, had none)
    static constexpr int ZOFF = (NSA * ASZ + NSB * BSZ + 2047) / 2048 * 2048;
    static constexpr int DUMP = ZOFF + 2048;                 // 1 KB: where the (NLOAD XLW - NPB) surplus pieces of a group land (every
                                                             // loader issues the same number of pieces: the vmcnt counts are constants)
    static constexpr int LDS = DUMP + 1024;
    static_assert(WC * WP == 8, "eight compute waves");
    static_assert(KS == 1 || KS == 2, "one or two K-halves per section");
    static_assert(NPA % NLOAD == 0 && CBW % 2 == 0, "weight pieces split evenly over the loaders; channel pairs per wave");
    static_assert(LDS <= 160 * 1024, "three weight tiles + two activation tiles must fit the LDS");
    static_assert(XLW <= 20, "row masks of the activation pieces are packed 3 bits each, ten per register, two registers");
};

#if defined(__HIP_DEVICE_COMPILE__)
// The tile list of a block: [tile0, tend) of its XCD's contiguous range, stride = blocks per XCD (every wave computes the same).
struct D3Tiles {
    int tile0, tend, tstride, nMine;
};
__device__ __forceinline__ D3Tiles d3q_tiles(const D3Params& p) {
    D3Tiles t;
    const int perXcd = (p.nTiles + 7) >> 3;
    const int xcd = blockIdx.x & 7;
    t.tstride = gridDim.x >> 3;                               // host guarantees gridDim.x % 8 == 0
    t.tile0 = xcd * perXcd + (int)(blockIdx.x >> 3);
    t.tend = min(p.nTiles, (xcd + 1) * perXcd);
    t.nMine = t.tile0 < t.tend ? (t.tend - t.tile0 + t.tstride - 1) / t.tstride : 0;
    return t;
}
__device__ __forceinline__ void d3q_sync() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

// ---- loader wave lw (0..3): every LDS-DMA piece of the block ------------------------------------------------------------------------
template <int DT, int WC, int WP, int CBW, int PBW, int KS, bool ONE>
__device__ __forceinline__ void d3q_loader(const D3Params& p, char* smem, const int lw) {
    typedef D3Cfg<WC, WP, CBW, PBW, KS, ONE> G;
    constexpr int BM = G::BM, BP = G::BP, WLW = G::WLW, XLW = G::XLW, NSA = G::NSA, NL = G::NLOAD;
    typedef __attribute__((address_space(3))) char lds_char;
    const int lane = threadIdx.x & 63;
    const int lrow = lane >> 3;
    const int cs = (lane & 7) ^ lrow;                         // K-chunk this lane fetches (source-side swizzle)
    const uint32_t lds0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(size_t)PCV_LDS(smem));
    const D3Tiles T = d3q_tiles(p);
    if (T.nMine == 0) return;
    const int nk = p.nk;
    const int K_total = T.nMine * nk, G_total = ONE ? K_total : T.nMine * (nk / 3);

    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);
    // Activations: the descriptor's base sits one image row BELOW x, so that the (wave-uniform, unsigned) scalar offset of a
    // group, (r * W * Cin + 64 c) elements, reaches the row above a pixel with r = 0. num_records covers the per-lane offset
    // of any pixel plus the largest scalar offset (whether or not the range check adds the scalar offset, a valid lane passes
    // it); a row that must read as zeros gets the offset 2^31, beyond num_records either way (the host keeps
    // x_bytes + 2 rows below 2^31).
    const uint32_t rowBytes = ONE ? 0u : (uint32_t)(p.W * p.Cin * 2);
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.x)) - rowBytes, 0, p.x_bytes + 2u * rowBytes, 0x00020000);

    // weights: K-step `la_k` of tile `la_tile` goes to A slot la_slot; loader lw owns pieces NL i + lw (rows 8 (NL i + lw) + lrow)
    int la_tile = T.tile0, la_k = 0, la_slot = 0, la_g = 0;    // la_g: global index of the next K-step to issue
    uint32_t woff0 = 0;
    auto setup_a = [&](int t) __attribute__((always_inline)) {
        const int chTile = t % p.nChTiles;
        woff0 = (uint32_t)(((chTile * BM + 8 * lw + lrow) * p.Kpad + cs * 8) * 2);       // rows past the blob: out of range -> zeros
    };
    const uint32_t wstep = (uint32_t)(8 * NL * p.Kpad * 2);    // NL pieces x 8 rows further down the blob
    auto dma_a = [&](auto I0c, auto I1c) __attribute__((always_inline)) {                // pieces [I0, I1) of the weight tile
        constexpr int I0 = decltype(I0c)::value, I1 = decltype(I1c)::value;
#pragma unroll
        for (int i = I0; i < I1; ++i) {
            const uint32_t dst = lds0 + (uint32_t)(la_slot * G::ASZ + (NL * i + lw) * 1024);
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
    // activations: group (lb_r, lb_c) of tile `lb_tile` goes to B slot lb_slot; loader lw owns pieces NL j + lw: LDS rows
    // u = 8 (NL j + lw) + lrow <-> flat pixel P0 + u - 1, input pixel = that pixel shifted by (lb_r - 1) image rows
    int lb_tile = T.tile0, lb_r = 0, lb_c = 0, lb_slot = 0, lb_g = 0;
    uint32_t pbv[XLW];             // byte offset of the pixel itself (+ this lane's chunk), or 2^31 for a row outside [0, M) / the tile
    uint32_t vmask[2] = {0u, 0u};  // 3 bits per piece (ten per register): image row ho + r - 1 exists, r = 0, 1, 2
    // The table of the NEXT tile is built ahead of the tile change, a third per K-step (`table_step`), into its own registers:
    // built at the change itself (15 pieces x ~20 VALU in the loader that is just then due to issue the next group) it delayed
    // every barrier of that K-step - 10-13 us of the 93 us of the 56x56x64 layers (9 K-steps per tile), 3.4 us at 28x28x128.
    uint32_t pbvN[XLW];
    uint32_t vmaskN[2] = {0u, 0u};
    int prep_t = -1, prep_c = 3;   // tile whose table pbvN / vmaskN hold or are building; thirds done (3 = complete)
    auto table_rows = [&](int t, auto J0c, auto J1c, uint32_t (&pb)[XLW], uint32_t (&vmk)[2]) __attribute__((always_inline)) {
        constexpr int J0 = decltype(J0c)::value, J1 = decltype(J1c)::value;
        const int tileP0 = (t / p.nChTiles) * BP;
#pragma unroll
        for (int j = J0; j < J1; ++j) {
            const int u = 8 * (NL * j + lw) + lrow;
            const int m = tileP0 + u - 1;
            uint32_t off = 0x80000000u, vm = 0;
            if (u < BP + 2 && m >= 0 && m < p.M) {
                const uint32_t n = fastdiv((uint32_t)m, p.div_hw);
                const uint32_t rem = (uint32_t)m - n * (uint32_t)p.HW;               // pixel index inside its image
                if constexpr (ONE) {
                    const uint32_t ho = fastdiv(rem, p.div_w);
                    const uint32_t wo = rem - ho * (uint32_t)p.W;
                    const uint32_t mi = (n * (uint32_t)p.Hin + ho * (uint32_t)p.stride) * (uint32_t)p.Win + wo * (uint32_t)p.stride;
                    off = (uint32_t)((mi * (uint32_t)p.Cin + (uint32_t)cs * 8u) * 2u);
                    vm = 7u;
                } else {
                    off = (uint32_t)((m * p.Cin + cs * 8) * 2);
                    // the rows above / below exist unless the pixel sits in the first / last image row: no second division
                    vm = (rem >= (uint32_t)p.W ? 1u : 0u) | 2u | (rem + (uint32_t)p.W < (uint32_t)p.HW ? 4u : 0u);
                }
            }
            pb[j] = off;
            vmk[j / 10] = (vmk[j / 10] & ~(7u << (3 * (j % 10)))) | (vm << (3 * (j % 10)));
        }
    };
    typedef std::integral_constant<int, 0> T0;
    typedef std::integral_constant<int, (XLW + 2) / 3> T1;
    typedef std::integral_constant<int, (2 * XLW + 2) / 3> T2;
    typedef std::integral_constant<int, XLW> T3;
    auto setup_b = [&](int t) __attribute__((always_inline)) { table_rows(t, T0{}, T3{}, pbv, vmask); };
    // One third of the next tile's table per K-step, thirds tied to the K-step's compile-time phase PH (3x3 mode: the filter column)
    // so that every table register is written at ONE place (with a run-time third the compiler merged the three branches into an
    // indexed store - the table went to scratch): third 0 at phase 2 (the K-step behind a tile change, which happens at phase 1),
    // third 1 at phase 0, third 2 at phase 1.
    auto table_step = [&](auto PHc) __attribute__((always_inline)) {
        constexpr int PH = decltype(PHc)::value;
        if (prep_t < 0) return;
        if constexpr (PH == 2) { if (prep_c == 0) { table_rows(prep_t, T0{}, T1{}, pbvN, vmaskN); prep_c = 1; } }
        if constexpr (PH == 0) { if (prep_c == 1) { table_rows(prep_t, T1{}, T2{}, pbvN, vmaskN); prep_c = 2; } }
        if constexpr (PH == 1) { if (prep_c == 2) { table_rows(prep_t, T2{}, T3{}, pbvN, vmaskN); prep_c = 3; } }
    };
    auto next_table = [&](int t) __attribute__((always_inline)) {   // tile t becomes the loaders' current tile
        if (prep_t == t && prep_c == 3 && !D3_DBG(16)) {
#pragma unroll
            for (int j = 0; j < XLW; ++j) pbv[j] = pbvN[j];
            vmask[0] = vmaskN[0];
            vmask[1] = vmaskN[1];
        } else {
            setup_b(t);                                         // tiles shorter than three K-steps: built on the spot
        }
        prep_t = t + T.tstride < T.tend ? t + T.tstride : -1;
        prep_c = 0;
    };
    auto dma_b = [&](auto J0c, auto J1c) __attribute__((always_inline)) {                // pieces [J0, J1) of the group
        constexpr int J0 = decltype(J0c)::value, J1 = decltype(J1c)::value;
        const uint32_t soff = ONE ? (uint32_t)(lb_c * 128) : (uint32_t)(lb_r * p.W * p.Cin + lb_c * 64) * 2u;
#pragma unroll
        for (int j = J0; j < J1; ++j) {
            const uint32_t dst = lds0 + (uint32_t)(NL * j + lw < G::NPB ? NSA * G::ASZ + lb_slot * G::BSZ + (NL * j + lw) * 1024 : G::DUMP);
            const uint32_t t = (uint32_t)__builtin_amdgcn_sbfe((int)vmask[j / 10], 3 * (j % 10) + lb_r, 1);    // all ones: the image row exists
            const uint32_t voff = (t & pbv[j]) | (~t & 0x80000000u);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_char*)(size_t)dst, 16, voff, soff, 0, 0);
        }
    };
    auto advance_b = [&]() __attribute__((always_inline)) {   // group order inside a tile: (r, c), c fastest
        ++lb_g;
        lb_slot = lb_slot + 1 == G::NSB ? 0 : lb_slot + 1;
        if (++lb_c == p.slices) {
            lb_c = 0;
            if (ONE || ++lb_r == 3) {
                lb_r = 0;
                lb_tile += T.tstride;
                if (lb_tile < T.tend && !D3_DBG(4)) next_table(lb_tile);
            }
        }
    };
    typedef std::integral_constant<int, 0> C0;
    typedef std::integral_constant<int, G::NB0> CB0;
    typedef std::integral_constant<int, XLW> CBN;
    constexpr int WA0 = (WLW + 1) / 2;                          // weight pieces issued in the first half of a K-step
    typedef std::integral_constant<int, WA0> CA0;
    typedef std::integral_constant<int, WLW> CAN;

    // prologue: weight tiles of K-steps 0 and 1, activation tile of group 0, the zero row
    setup_a(T.tile0);
    setup_b(T.tile0);
    prep_t = T.tile0 + T.tstride < T.tend ? T.tile0 + T.tstride : -1;
    prep_c = 0;
    if (lw < 2) *reinterpret_cast<__attribute__((address_space(3))) u32x4*>((size_t)(lds0 + G::ZOFF + (lw * 64 + lane) * 16)) = (u32x4){0u, 0u, 0u, 0u};
    dma_b(C0{}, CBN{});
    advance_b();
    dma_a(C0{}, CAN{});
    advance_a();
    if (K_total > 1) {
        dma_a(C0{}, CAN{});
        advance_a();
        if constexpr (ONE) {                                    // 1x1: the activation tile of K-step 1 too
            dma_b(C0{}, CBN{});
            advance_b();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    d3q_sync();

    // One K-step: the pieces of K-step s + 2 (weights) and, during q = 0 and q = 1, of the next group (activations), spread
    // over the K-step's barrier intervals; before its last barrier everything issued BEFORE this K-step has landed - the
    // weight tile of K-step s + 1 and, behind q = 2, the activation tile of the next group.
    auto kstep = [&](auto Qc) __attribute__((always_inline)) {
        constexpr int Q = decltype(Qc)::value;
        constexpr int NBQ = Q == 0 ? G::NB0 : (Q == 1 ? G::NB1 : 0);
        const bool moreA = la_g < K_total, moreB = lb_g < G_total;      // K-step s + 2 / group g + 1 exist
        if constexpr (Q == 0) { if (moreB) dma_b(C0{}, CB0{}); }
        if constexpr (Q == 1) { if (moreB) { dma_b(CB0{}, CBN{}); advance_b(); } }
        if (moreA) dma_a(C0{}, CA0{});
        table_step(Qc);                                         // (behind this interval's pieces, in front of its barrier)
        d3q_sync();
        if constexpr (KS == 1) d3q_sync();
        if (moreA) {
            dma_a(CA0{}, CAN{});
            advance_a();
        }
        if constexpr (KS == 1) d3q_sync();
        if (moreA && (NBQ == 0 || moreB)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WLW + NBQ) : "memory");
        else if (moreA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WLW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        d3q_sync();
    };
    if constexpr (!ONE) {
        for (int s = 0; s < K_total; s += 3) {
            kstep(std::integral_constant<int, 0>{});
            kstep(std::integral_constant<int, 1>{});
            kstep(std::integral_constant<int, 2>{});
        }
    } else {
        // 1x1: K-step s issues the weight AND the activation tile of K-step s + 2, half of each per half K-step; before its last
        // barrier everything issued before this K-step (both tiles of K-step s + 1) has landed.
        for (int s = 0; s < K_total; ++s) {
            const bool moreA = la_g < K_total, moreB = lb_g < G_total;          // (always equal: one group per K-step)
            if (moreB) dma_b(C0{}, CB0{});
            if (moreA) dma_a(C0{}, CA0{});
            d3q_sync();
            if constexpr (KS == 1) d3q_sync();
            if (moreB) {
                dma_b(CB0{}, CBN{});
                advance_b();
            }
            if (moreA) {
                dma_a(CA0{}, CAN{});
                advance_a();
            }
            if constexpr (KS == 1) d3q_sync();
            if (moreA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WLW + XLW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            d3q_sync();
        }
    }
}

// ---- compute waves: the whole persistent loop of one group (GRP 0: waves 0-3, GRP 1: waves 4-7, one barrier interval behind). ------
// The two instantiations are separate straight-line loop nests (no per-interval group branches for the register allocator to join).
template <int DT, int WC, int WP, int CBW, int PBW, int KS, int GRP, bool ONE>
__device__ __forceinline__ void d3q_body(const D3Params& p, char* smem, const int wave) {
    typedef D3Cfg<WC, WP, CBW, PBW, KS, ONE> G;
    constexpr int BM = G::BM, BP = G::BP, NSA = G::NSA;
    typedef typename Mma<DT>::frag frag;

    const int lane = threadIdx.x & 63;
    const int wc = wave / WP, wp = wave % WP;
    const int fr = lane & 15, fq = lane >> 4;
    const uint32_t lds0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(size_t)PCV_LDS(smem));
    const D3Tiles T = d3q_tiles(p);
    if (T.nMine == 0) return;
    const int nk = p.nk;
    const int K_total = T.nMine * nk;                         // K-steps this block walks

    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res), 0, p.res != nullptr ? p.res_bytes : 0u, 0x00020000);

    f32x4 acc[CBW][PBW];
    frag a[KS][CBW], b[KS][PBW];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < CBW; ++i)
#pragma unroll
            for (int j = 0; j < PBW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    };
    // fragment addresses: A row = wave's channel row, B row = wave's pixel row + 1 + (q - 1); the swizzle key is the row & 7
    const uint32_t afrag = lds0 + (uint32_t)((wc * 16 * CBW + fr) * 128);
    const uint32_t brow0 = (uint32_t)(wp * 16 * PBW + fr);
    uint32_t hm0 = 0, hm2 = 0;     // bit j: this lane's output pixel of block j is in image column 0 / W - 1 (set per tile)
    uint32_t hmN = 0;              // the same two masks of the NEXT tile (hm0 in bits 0-15, hm2 in bits 16-31), computed during this tile's
                                   // last K-step beside the BN prefetch: at the tile change itself it stood between two tiles' K loops
    auto masks_of = [&](int t) __attribute__((always_inline)) -> uint32_t {
        const int m0 = (t / p.nChTiles) * BP + wp * 16 * PBW + fr;
        uint32_t h = 0;
#pragma unroll
        for (int j = 0; j < PBW; ++j) {
            const uint32_t m = (uint32_t)(m0 + 16 * j);
            const uint32_t wo = m - fastdiv(m, p.div_w) * (uint32_t)p.W;              // (n H + ho) W + wo = m
            h |= (wo == 0u ? 1u : 0u) << j;
            h |= (wo + 1u == (uint32_t)p.W ? 1u : 0u) << (16 + j);
        }
        return h;
    };
    auto setup_masks = [&](int t) __attribute__((always_inline)) {
        const uint32_t h = masks_of(t);
        hm0 = h & 0xFFFFu;
        hm2 = h >> 16;
    };
    // section h (of 2 / KS) of K-step (A slot sa, B slot sb, filter column Q): K-halves h * KS .. h * KS + KS - 1.
    // Fragment i / j sits i / j * 2048 bytes behind the wave's first row: pointer arithmetic, so that the constant folds into the
    // ds_read offset field. A horizontally padded tap selects the BASE (zero row - j * 2048) per lane, the offset stays immediate.
    typedef const __attribute__((address_space(3))) char* lds_cptr;
    typedef const __attribute__((address_space(3))) frag* lds_fptr;
    auto reads = [&](int sa, int sb, auto Qc, int h) __attribute__((always_inline)) {
        constexpr int Q = decltype(Qc)::value;
        const uint32_t abase = afrag + (uint32_t)(sa * G::ASZ);
        const uint32_t brow = brow0 + Q;
        const uint32_t bbase = lds0 + (uint32_t)(NSA * G::ASZ + sb * G::BSZ) + brow * 128u;
        const uint32_t zrow = lds0 + (uint32_t)G::ZOFF;
#pragma unroll
        for (int u = 0; u < KS; ++u) {
            const uint32_t kc = (uint32_t)(fq + 4 * (h * KS + u));
            lds_cptr ap = (lds_cptr)(size_t)(abase + ((kc ^ (uint32_t)(fr & 7)) << 4));
            lds_cptr bp = (lds_cptr)(size_t)(bbase + ((kc ^ (brow & 7u)) << 4));
            const uint32_t zsel = zrow + ((uint32_t)(size_t)bp & 2047u);        // the zero block through this lane's own banks
#pragma unroll
            for (int i = 0; i < CBW; ++i) a[u][i] = *reinterpret_cast<lds_fptr>(ap + i * 2048);
#pragma unroll
            for (int j = 0; j < PBW; ++j) {
                lds_cptr bj = bp;
                if constexpr (Q != 1) {
                    const uint32_t t = (uint32_t)__builtin_amdgcn_sbfe((int)(Q == 0 ? hm0 : hm2), j, 1);   // all ones: horizontally padded tap
                    bj = (lds_cptr)(size_t)((t & (zsel - (uint32_t)(j * 2048))) | (~t & (uint32_t)(size_t)bp));
                }
                b[u][j] = *reinterpret_cast<lds_fptr>(bj + j * 2048);
            }
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
    // The BN scale / shift of the wave's channels are requested at the START of the tile's last K-step, and a convolution WITHOUT skip
    // connection no longer issues residual loads at all (zero-record buffer loads return zeros but still take the memory pipeline's
    // ~1 us; fetched inside the epilogue these latencies stood between two tiles' K loops: 2 590 cycles per K-step on the 9-K-step
    // tiles of the 64-channel layers against ~1 500 elsewhere). The residual tile itself stays in the epilogue (28 registers).
    const bool has_res = p.res != nullptr;
    f32x4 es0[CBW / 2], es1[CBW / 2], eh0[CBW / 2], eh1[CBW / 2];
    auto ep_prefetch = [&](int t) __attribute__((always_inline)) {
        const int chTile = t % p.nChTiles;
#pragma unroll
        for (int ip = 0; ip < CBW / 2; ++ip) {
            const int ch0 = chTile * BM + wc * 16 * CBW + 32 * ip + 8 * fq;
            const int chl = ch0 < p.Cout ? ch0 : 0;              // table index of a pad channel: any valid one (never stored)
            es0[ip] = *reinterpret_cast<const f32x4*>(p.scale + chl); es1[ip] = *reinterpret_cast<const f32x4*>(p.scale + chl + 4);
            eh0[ip] = *reinterpret_cast<const f32x4*>(p.shift + chl); eh1[ip] = *reinterpret_cast<const f32x4*>(p.shift + chl + 4);
        }
    };
    auto epilogue = [&](int t) __attribute__((always_inline)) {
        if (D3_DBG(2)) return;
        const int chTile = t % p.nChTiles;
        const int tileP0 = (t / p.nChTiles) * BP;
        const int mBase = tileP0 + wp * 16 * PBW + fr;
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
                const bool ok = chok && m < p.M && !D3_DBG(1);
                const uint32_t boff = ok ? (uint32_t)(((size_t)m * p.Ypitch + ch0) * 2) : 0x80000000u;
                __builtin_amdgcn_raw_buffer_store_b128(o, yrsrc, boff, 0, 0);
            }
        }
        guard.commit(p.ovf);
    };

    zero_acc();
    if constexpr (!ONE) setup_masks(T.tile0);
    d3q_sync();                                                // the loaders' prologue: K-steps 0 and 1, group 0, the zero row

    int sa = 0, sb = 0, k = 0, cur_tile = T.tile0, ep_tile = T.tile0;
    bool ep = false;
#ifdef D3X3_STAMPS
    int stamps = 0;
#endif
    // One K-step (filter column Q of the current group). `s` = global K-step index of this block.
    //   interval 0: group 0 reads (first section) | group 1 finishes K-step s - 1 (+ epilogue)
    //   interval 1: group 0 computes | group 1 reads;  KS == 1: two more intervals for the second K-half
    // Returns true after the tail (s == K_total: group 1's last section and both groups' last epilogue).
    auto kstep = [&](int s, auto Qc) __attribute__((always_inline)) -> bool {
        constexpr int Q = decltype(Qc)::value;
        if constexpr (GRP == 1) {
            if (s > 0) mfmas();
        }
        if constexpr (ONE || Q == 0) {                        // a tile ends behind q = 2 (nk is a multiple of 3); 1x1: behind any K-step
            if (ep) {
                epilogue(ep_tile);
                zero_acc();
                if constexpr (!ONE) {
                    if (!D3_DBG(8)) {
                        hm0 = hmN & 0xFFFFu;
                        hm2 = hmN >> 16;
                    }
                }
            }
            if (s == K_total) return true;
        }
        if (k == nk - 1) {
            ep_prefetch(cur_tile);
            if constexpr (!ONE) hmN = masks_of(cur_tile + T.tstride);               // (past the last tile: computed, never used)
        }
        if constexpr (GRP == 0) {
            reads(sa, sb, Qc, 0);
            reads_done();
        }
        D3_STAMP(0);
        d3q_sync();
        D3_STAMP(1);
        if constexpr (GRP == 0) {
            mfmas();
        } else {
            reads(sa, sb, Qc, 0);
            reads_done();
        }
        if constexpr (KS == 1) {
            D3_STAMP(2);
            d3q_sync();
            D3_STAMP(3);
            if constexpr (GRP == 0) {
                reads(sa, sb, Qc, 1);
                reads_done();
            } else {
                mfmas();
            }
            D3_STAMP(4);
            d3q_sync();
            D3_STAMP(5);
            if constexpr (GRP == 0) {
                mfmas();
            } else {
                reads(sa, sb, Qc, 1);
                reads_done();
            }
        }
        ep = (k == nk - 1);
        ep_tile = cur_tile;
        if (++k == nk) {
            k = 0;
            cur_tile += T.tstride;
        }
        sa = sa + 1 == NSA ? 0 : sa + 1;
        if constexpr (ONE) sb = sb + 1 == G::NSB ? 0 : sb + 1;
        else if constexpr (Q == 2) sb ^= 1;
        D3_STAMP(7);
        d3q_sync();
        D3_STAMP(8);
        return false;
    };
#ifdef D3Q_CYCLES                                              // diagnostic build: shader cycles / real time of this block's K loop (tests/tools/d3q_cycles.py)
    const uint64_t cyc0__ = __builtin_amdgcn_s_memtime(), rt0__ = __builtin_amdgcn_s_memrealtime();
#endif
    if constexpr (ONE) {
        for (int s = 0;; ++s)
            if (kstep(s, std::integral_constant<int, 1>{})) break;        // the unmasked centre column: tile row u holds pixel P0 + u - 1
    } else {
        for (int s = 0;; s += 3) {
            if (kstep(s, std::integral_constant<int, 0>{})) break;
            if (kstep(s + 1, std::integral_constant<int, 1>{})) break;
            if (kstep(s + 2, std::integral_constant<int, 2>{})) break;
        }
    }
#ifdef D3X3_STAMPS
    if (p.dbg != nullptr && blockIdx.x == 16) p.dbg[wave * 64 + lane] = (uint32_t)stamps;
#endif
#ifdef D3Q_CYCLES
    if (p.dbg != nullptr && blockIdx.x == 16 && wave == 0 && lane == 0) {
        p.dbg[0] = (uint32_t)(__builtin_amdgcn_s_memtime() - cyc0__);
        p.dbg[1] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - rt0__);
        p.dbg[2] = (uint32_t)K_total;
    }
#endif
}
#endif  // __HIP_DEVICE_COMPILE__

template <int DT, int WC, int WP, int CBW, int PBW, int KS, bool ONE = false>
__global__ __launch_bounds__(768, 3) void d3q_kernel(const D3Params p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // compute waves w and w + 4 share a SIMD (a workgroup's waves are dealt to the SIMDs cyclically), loader w + 8 joins them
    if (wave < 4) d3q_body<DT, WC, WP, CBW, PBW, KS, 0, ONE>(p, smem, wave);
    else if (wave < 8) d3q_body<DT, WC, WP, CBW, PBW, KS, 1, ONE>(p, smem, wave);
    else d3q_loader<DT, WC, WP, CBW, PBW, KS, ONE>(p, smem, wave - 8);
#endif  // __HIP_DEVICE_COMPILE__
}

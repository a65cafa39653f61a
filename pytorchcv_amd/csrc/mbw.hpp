// mbw.hpp - the fused inverted-residual unit (mbconv.hpp: 1x1 expand + BN + act -> depthwise 3x3 + BN + act -> 1x1 project + BN
// (+ skip); reference LinearBottleneck.forward, mobilenetv2.py:62-71) with WAVE-PRIVATE tiles: no block barrier after the weights
// are in LDS.
//
// mbconv.hpp's block owns a TH x 16 output tile and its four waves meet at two barriers per 32-channel chunk; with 5 pixel blocks
// of expand work per wave between barriers the unit ran at 7-8 K cycles per chunk for ~1.5 K cycles of instructions (16 -> 96 ->
// 24 at 112x112: 559 us against a 70 us HBM floor). Here every wave owns a tile of 4 (stride 1) or 2 (stride 2) MFMA pixel blocks
// - a pixel block is 1 row x 16 columns (TW = 16) or 2 rows x 8 columns (TW = 8: a 56-wide map is 7 x 8 with no ragged column
// tile, and the stride-2 window shrinks from 5 x 33 to 9 x 17 pixels) - and keeps its own E / D chunk tiles in LDS, so its
// three stages are one straight instruction stream (LDS operations of a wave execute in order: no barrier, no explicit wait),
// and eight such waves per CU hide each other's latencies. The price is the expand work of the halo rows that neighbouring waves
// no longer share (1.6-1.7 / 4.8-5.2 expanded pixels per output at stride 1 / 2 instead of 1.4 / 4.6).
//
//   x   the window's pixels go global -> registers as MFMA B fragments (lane = pixel l % 16, channels 8 (l / 16) .. + 7; padding and
//       channel tails = out-of-range buffer offsets = zeros), prefetched one tile ahead. No LDS staging: Cin <= 32 is one K step.
//   S1  E[window pixel][32 ch] = act(BN(W_exp[chunk] . x)); outside the image forced to 0 (the depthwise pads the EXPANDED map)
//   S2  D[R x 16 pixels][32 ch] = act(BN(depthwise 3x3 of E)) as block-diagonal MFMAs (2 taps x 16 channels per K = 32 step, 5 steps)
//   S3  acc[Cout][R x 16] += W_proj[:, chunk] . D
//   then BN (+ residual) and 16-byte NHWC stores.
// All LDS rows (weights, E, D) are 64 bytes = four 16-byte slots; slot s of row r is stored at slot s ^ (2 * bit 2 of r). With
// ds_read_b128's real lane groups ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... - MI355X_MICROARCH.md, LDS) this swizzle is
// conflict-free for 16 CONSECUTIVE rows at ANY base row (brute-forced over all bases), which is what every fragment access here
// is: weight rows, the pixels of a D block, and the 16 pixels of one output row under one tap - at stride 2 because the window is
// stored with every row split into its even and its odd columns. (Round 2's first version used unswizzled rows on an 80-byte
// pitch, derived from contiguous 16-lane groups: SQ_LDS_BANK_CONFLICT showed 2.1-4.7 extra cycles on EVERY LDS instruction and the
// LDS busy 47-67 % of the kernel's time.) The 2 x 8 pixel blocks read two runs of 8 rows and stay 2-4-way conflicted unless the
// runs are 8 k rows apart; they are kept selectable, the default is 1 x 16.
#pragma once
#include "pcv_common.hpp"
#include "igemm_conv.hpp"     // Mma<DT>
#include "mbconv.hpp"         // MbParams

struct MbwLds {
    int wexp, wproj, wdw, bn, wave0, per_wave, total;
};
// 16-pixel blocks over the input window of a wave tile
// rb: pixel blocks per wave tile (0 = the default: 4 at stride 1, 2 at stride 2)
static inline __host__ __device__ int mbw_npt(int stride, int tw, int rb = 0) {
    const int nblk = rb > 0 ? rb : (stride == 1 ? 4 : 2), ro = nblk * (16 / tw);
    const int iw = (tw - 1) * stride + 3, iws = iw + ((stride == 2 && tw == 8) ? 1 : 0);      // storage pitch of a window row
    return (((ro - 1) * stride + 3) * iws + 15) / 16;                                        // 7 / 7 (stride 1), 11 / 11 (stride 2)
}
// ka: K steps of the expand GEMM (Cin <= 32 ka). With ka == 1 the two 1x1 weight matrices live in LDS; wider units (64 -> 384 ->
// 64 is 98 KB of weights) leave them in L2 and every wave fetches its fragments per chunk.
static inline __host__ __device__ MbwLds mbw_lds_layout(int stride, int nrt, int nChunks, int nWaves, int tw, int ka = 1, int rb = 0) {
    const int npt = mbw_npt(stride, tw, rb), rows = rb > 0 ? rb : (stride == 1 ? 4 : 2);
    MbwLds L;
    int o = 0;
    const int pitch = ka == 1 ? 64 : 80;                          // (the two-K-step variant keeps unswizzled rows on an 80-byte pitch: below)
    L.wexp = o; o += ka == 1 ? nChunks * 32 * 64 : 0;             // [chunk][32 rows]
    L.wproj = o; o += ka == 1 ? nChunks * nrt * 16 * 64 : 0;      // [chunk][nrt * 16 rows]
    L.wdw = o; o += (10 * nChunks * 32 * 2 + 15) & ~15;           // [10][CmidP] 16-bit: tap 9 = zeros (second half of the last tap pair)
    L.bn = o; o += 4 * nChunks * 32 * 4;                          // scale_e, shift_e, scale_d, shift_d
    L.wave0 = o;
    L.per_wave = (npt * 16 + rows * 16) * pitch;                  // E tile + D tile
    o += nWaves * L.per_wave;
    L.total = o;
    return L;
}

template <bool SWZ> __device__ __forceinline__ int mbw_swz(int row) { return SWZ ? (row >> 1) & 2 : 0; }      // slot XOR of a 64-byte LDS row

// The activation behind the expand and the depthwise stage as a compile-time constant: with the launch-time code every one of the
// 11 + 4 unrolled applications per chunk was a nest of scalar branches (580 basic blocks), and nothing could be scheduled across them.
template <int ACT, int N> __device__ __forceinline__ void mbw_act(float (&v)[N], const ActClamp& dyn) {
    if constexpr (ACT == PCV_ACT_RELU) {
#pragma unroll
        for (int e = 0; e < N; ++e) v[e] = __builtin_elementwise_maximum(v[e], 0.f);
    } else if constexpr (ACT == PCV_ACT_RELU6) {
#pragma unroll
        for (int e = 0; e < N; ++e) v[e] = __builtin_elementwise_minimum(__builtin_elementwise_maximum(v[e], 0.f), 6.f);
    } else {
        apply_actn<N>(v, dyn);
    }
}

// S: stride; NRT: 16-row tiles of the project GEMM (2: Cout <= 32, 4: Cout <= 64); ACT: activation of the expand and depthwise
// stages when both are ReLU or both ReLU6, -1 = read p.act_e / p.act_d; TW: columns of a pixel block (16 or 8). Cin <= 32 KA.
// RB: pixel blocks per wave tile, 0 = 4 (stride 1) / 2 (stride 2). The WIDE units (96 projected channels = NRT 6, up to three expand
// K steps: MobileNetV2 units 11-13, 64 / 96 -> 384 / 576 -> 96 at 14x14) run with RB = 2: 48 instead of 96 accumulator registers and
// 5 instead of 7 window blocks of x fragments per K step keep the wave inside 256 registers (one 8-wave block per CU).
// blockDim.x = 64 * waves.
template <int DT, int S, int NRT, int ACT, int TW, int KA = 1, int RB = 0>
__global__ __launch_bounds__(512) void mbw_kernel(const MbParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int R = RB > 0 ? RB : (S == 1 ? 4 : 2);   // pixel blocks per wave tile
    constexpr int PR = 16 / TW;                         // output rows of a pixel block
    constexpr int RO = R * PR;                          // output rows of the tile
    constexpr int IH = (RO - 1) * S + 3, IW = (TW - 1) * S + 3;
    // Storage pitch of a window row. A 2 x 8 pixel block reads two runs of 8 rows: conflict-free when the runs are 8 k rows apart,
    // so its two output rows are R apart (stride 1: 4 x 10 = 40 storage rows) and the stride-2 window row is padded 17 -> 18
    // (4 x 18 = 72); the 1 x 16 blocks read 16 consecutive rows and need neither.
    constexpr int IWS = IW + ((S == 2 && TW == 8) ? 1 : 0);
    constexpr int NIP = IH * IWS;                       // stored window pixels: 108 / 100 (stride 1), 165 / 162 (stride 2)
    constexpr int NPT = (NIP + 15) / 16;
    // The two-K-step variant has no registers for a table of swizzled S2 addresses (and computing them per read cost 24 %): it keeps
    // round 2's first layout - unswizzled rows on an 80-byte pitch, every S2 address an immediate off 5 registers, 2.7 conflict cycles
    // per LDS instruction - which measured faster there (51.5 vs 63.9 us).
    constexpr bool SWZ = KA == 1;
    constexpr int PITCH = SWZ ? 64 : 80;
    constexpr int IWH = (IW + 1) / 2;                    // stride 2: a window row is stored as its even columns, then its odd columns
    typedef typename Mma<DT>::frag frag;
    // VALU instructions of one S1 pixel block between two MFMA groups (BN 4 packed fma, 16 clamps, 4 packs, address / select) - the
    // fp16 ReLU build checks the rounded range on top (F16Guard: 4 x v_max(3)_f32 + v_cmp)
    constexpr int S1VALU = (DT == PCV_F16 && ACT == PCV_ACT_RELU6) ? 14 : 26 + ((DT == PCV_F16 && ACT == PCV_ACT_RELU) ? 5 : 0);   // (4 pk_fma + 4 cvt + 8 packed clamps)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nWaves = blockDim.x >> 6;
    constexpr bool WLDS = KA == 1;                      // 1x1 weights resident in LDS (else: fragments straight from L2)
    const MbwLds L = mbw_lds_layout(S, NRT, p.nChunks, nWaves, TW, KA, RB);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    char* const Wes = smem + L.wexp;
    char* const Wps = smem + L.wproj;
    char* const Wds = smem + L.wdw;
    float* const BNs = reinterpret_cast<float*>(smem + L.bn);
    char* const Es = smem + L.wave0 + wave * L.per_wave;
    char* const Ds = Es + NPT * 16 * PITCH;
    const int CmidP = p.nChunks * 32;

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res), 0, p.res ? p.y_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t wersrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w_exp), 0, p.wexp_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wdrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w_dw), 0, p.wdw_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wprsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w_proj), 0, p.wproj_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t sprsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.scale_p), 0, p.Cout * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t hprsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.shift_p), 0, p.Cout * 4, 0x00020000);

    // ---- the unit's weights -> LDS, once per block (rows beyond the packed matrices / channels beyond Cmid read as zeros) --------
    for (int i = tid; WLDS && i < p.nChunks * 32 * 4; i += blockDim.x) {
        const int slot = i & 3, row = i >> 2;                                      // row = 32 c + r: packed row order = MFMA order
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wersrc, (uint32_t)((row * p.Kpad1 + 8 * slot) * 2), 0, 0);
        *reinterpret_cast<u32x4*>(Wes + row * PITCH + ((slot ^ mbw_swz<SWZ>(row)) << 4)) = v;
    }
    for (int i = tid; WLDS && i < p.nChunks * NRT * 16 * 4; i += blockDim.x) {
        const int slot = i & 3, row = (i >> 2) % (NRT * 16), c = (i >> 2) / (NRT * 16);
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wprsrc, (uint32_t)((row * p.Kpad2 + 32 * c + 8 * slot) * 2), 0, 0);
        *reinterpret_cast<u32x4*>(Wps + (c * NRT * 16 + row) * PITCH + ((slot ^ mbw_swz<SWZ>(row)) << 4)) = v;
    }
    for (int i = tid; i < 10 * CmidP / 8; i += blockDim.x) {
        const int t = i / (CmidP / 8), ch = (i - t * (CmidP / 8)) * 8;
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wdrsrc, (t < 9 && ch < p.Cmid) ? (uint32_t)((t * p.Cmid + ch) * 2) : 0x80000000u, 0, 0);
        *reinterpret_cast<u32x4*>(Wds + (t * CmidP + ch) * 2) = v;
    }
    for (int i = tid; i < 4 * CmidP; i += blockDim.x) {
        const int which = i / CmidP, ch = i - which * CmidP;
        const float* src = which == 0 ? p.scale_e : which == 1 ? p.shift_e : which == 2 ? p.scale_d : p.shift_d;
        BNs[i] = (ch < p.Cmid && src != nullptr) ? src[ch] : 0.f;
    }
    __syncthreads();                                                               // the only barrier of the kernel

    const ActClamp act_e = make_act(p.act_e), act_d = make_act(p.act_d), act_p = make_act(p.act_p), post = make_act(p.post);

    // ---- lane constants ----------------------------------------------------------------------------------------------------------
    // window pixel 16 m + fr of this lane: (row << 8) | column; pixels past the window get row 255 (never inside an image: H <= 250)
    uint32_t prc[NPT];
#pragma unroll
    for (int m = 0; m < NPT; ++m) {
        const int ip = 16 * m + fr;                                                // storage index of the window pixel this lane expands
        const int wr = ip / IWS, cc = ip % IWS;
        const int wcol = S == 1 ? cc : (cc < IWH ? 2 * cc : 2 * (cc - IWH) + 1);
        prc[m] = (ip < NIP && wcol < IW) ? (uint32_t)((wr << 8) | wcol) : 0xFF00u;
    }
    // S2: A (weights, rows = channels) lane (row fr, k quarter fq) holds tap (fq >> 1) of the pair, channels 8 (fq & 1) .. + 7 of the
    // 16-channel half: non-zero only on the diagonal, element fr & 7 when (fr >> 3) == (fq & 1). B (E tile, columns = the 16 pixels
    // of one pixel block): lane (pixel fr, fq) reads the same 8 channels of the input pixel under its tap.
    const int pr = fr / TW, pc = fr % TW;                                          // this lane's pixel inside a pixel block
    uint32_t am[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) am[i] = ((fr >> 3) == (fq & 1) && i == ((fr & 7) >> 1)) ? 0xFFFFFFFFu : 0u;
    const int a_sh = (fr & 1) * 16;
    const char* const a_w = Wds + ((fq >> 1) * CmidP + fr) * 2;                    // + (2 j CmidP + 32 c + 16 g) * 2
    // S2 read addresses (channel half g = 0; g = 1 is the address ^ 32 = + 32 without the swizzle): one register per (pixel block,
    // tap pair) with the swizzle; affine in the pixel block without it (block 0's addresses + an immediate).
    constexpr bool BTAB = SWZ;
    uint32_t boff[BTAB ? R : 1][5];
#pragma unroll
    for (int u = 0; u < (BTAB ? R : 1); ++u)
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int tap = min(2 * j + (fq >> 1), 8);                             // tap 9 has zero weights: any valid address
            const int wr = (u + R * pr) * S + tap / 3, wcol = pc * S + tap % 3;       // pixel block u = output rows u (and u + R)
            const int sidx = wr * IWS + (S == 1 ? wcol : (wcol & 1) * IWH + (wcol >> 1));
            boff[u][j] = (uint32_t)(sidx * PITCH + (((fq & 1) ^ mbw_swz<SWZ>(sidx)) << 4));
        }
    const int fsw = (fq ^ mbw_swz<SWZ>(fr)) << 4;                                       // fragment access of row (16 k + fr), slot fq
    const char* const e_wr = Es + fr * PITCH + fsw;                                // S1 writes: + 16 m * PITCH
    const char* const d_rd = Ds + fr * PITCH + fsw;
    // D write: 8 bytes = channels 16 g + 4 fq .. + 3 of pixel (16 u + fr) = slot 2 g + (fq >> 1), half fq & 1; g = 1 is ^ 32
    char* const d_wr0 = Ds + fr * PITCH + ((((fq >> 1) ^ mbw_swz<SWZ>(fr))) << 4) + 8 * (fq & 1);
    char* const d_wr1 = Ds + fr * PITCH + ((((2 + (fq >> 1)) ^ mbw_swz<SWZ>(fr))) << 4) + 8 * (fq & 1);
    const char* const we_rd = Wes + fr * PITCH + fsw;
    const char* const wp_rd = Wps + fr * PITCH + fsw;

    const int nWavesAll = gridDim.x * nWaves;
    int tile = blockIdx.x * nWaves + wave;

    // ---- x fragments of tile t: global -> registers ---------------------------------------------------------------------------------
    auto load_x = [&](int t, u32x4 (&xr)[KA][NPT], uint32_t& vmask) __attribute__((always_inline)) {
        const int tw = t % p.tilesW;
        const int t2 = t / p.tilesW;
        const int th = t2 % p.tilesH;
        const int n = t2 / p.tilesH;
        const int hi0 = th * RO * S - 1, wi0 = tw * TW * S - 1;
        const bool live = t < p.nTiles;
        vmask = 0;
#pragma unroll
        for (int m = 0; m < NPT; ++m) {
            const int hi = hi0 + (int)(prc[m] >> 8), wi = wi0 + (int)(prc[m] & 255u);
            const bool ok = live & ((unsigned)hi < (unsigned)p.H) & ((unsigned)wi < (unsigned)p.W);      // (no short circuit: no branches)
            vmask |= ok ? (1u << m) : 0u;
            const uint32_t off = (uint32_t)((((n * p.H + hi) * p.W + wi) * p.Cin + 8 * fq) * 2);        // < 2 GiB: checked by the host
#pragma unroll
            for (int ks = 0; ks < KA; ++ks)
                xr[ks][m] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (ok & (32 * ks + 8 * fq < p.Cin)) ? off + 64 * ks : 0x80000000u, 0, 0);
        }
    };

    constexpr bool XPRE = KA == 1;                      // prefetch the next tile's x (register budget: only the one-K-step units)
    u32x4 xa[KA][NPT];
    uint32_t vm;
    if constexpr (XPRE) load_x(tile, xa, vm);

    while (tile < p.nTiles) {
        const int ntile = tile + nWavesAll;
        u32x4 xb[XPRE ? KA : 1][XPRE ? NPT : 1];
        uint32_t vmn = 0;
        if constexpr (XPRE) load_x(ntile, xb, vmn);                                // in flight during the whole tile
        else load_x(tile, xa, vm);

        const int tw = tile % p.tilesW;
        const int t2 = tile / p.tilesW;
        const int th = t2 % p.tilesH;
        const int n = t2 / p.tilesH;
        const int ho0 = th * RO, wo0 = tw * TW;

        f32x4 acc[NRT][R];
#pragma unroll
        for (int i = 0; i < NRT; ++i)
#pragma unroll
            for (int u = 0; u < R; ++u) acc[i][u] = (f32x4){0.f, 0.f, 0.f, 0.f};

        // weights from L2 (KA > 1): expand fragments one chunk ahead, projection fragments from the top of their chunk
        u32x4 wen[WLDS ? 1 : KA][2], wpn[WLDS ? 1 : NRT];
        auto load_we = [&](int c) __attribute__((always_inline)) {
#pragma unroll
            for (int ks = 0; ks < KA; ++ks)
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    wen[ks][i] = __builtin_amdgcn_raw_buffer_load_b128(wersrc, (uint32_t)(((32 * c + 16 * i + fr) * p.Kpad1 + 32 * ks + 8 * fq) * 2), 0, 0);
        };
        if constexpr (!WLDS) load_we(0);

#pragma unroll 1
        for (int c = 0; c < p.nChunks; ++c) {
            F16Guard<DT, true> g1, g2;                                // fp16 range checks of this chunk's E / D values (unbounded activations only)
            // ---- S1: E chunk over the whole window ---------------------------------------------------------------------------------
            {
                frag we[KA][2];
#pragma unroll
                for (int ks = 0; ks < KA; ++ks)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        if constexpr (WLDS) we[ks][i] = *reinterpret_cast<const frag*>(we_rd + (32 * c + 16 * i) * PITCH);
                        else we[ks][i] = __builtin_bit_cast(frag, wen[ks][i]);
                    }
                if constexpr (!WLDS) {                                             // next chunk's expand fragments + this chunk's projection
                    load_we(c + 1 < p.nChunks ? c + 1 : c);                        // fragments: in flight under S1 / S2
#pragma unroll
                    for (int i = 0; i < NRT; ++i)
                        wpn[i] = __builtin_amdgcn_raw_buffer_load_b128(wprsrc, (uint32_t)(((16 * i + fr) * p.Kpad2 + 32 * c + 8 * fq) * 2), 0, 0);
                }
                const f32x4 se0 = *reinterpret_cast<const f32x4*>(BNs + 32 * c + 8 * fq);
                const f32x4 se1 = *reinterpret_cast<const f32x4*>(BNs + 32 * c + 8 * fq + 4);
                const f32x4 he0 = *reinterpret_cast<const f32x4*>(BNs + CmidP + 32 * c + 8 * fq);
                const f32x4 he1 = *reinterpret_cast<const f32x4*>(BNs + CmidP + 32 * c + 8 * fq + 4);
#pragma unroll
                for (int m = 0; m < NPT; ++m) {
                    f32x4 e0 = (f32x4){0.f, 0.f, 0.f, 0.f}, e1 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < KA; ++ks) {
                        const frag b = __builtin_bit_cast(frag, xa[ks][m]);
                        e0 = Mma<DT>::run(we[ks][0], b, e0);
                        e1 = Mma<DT>::run(we[ks][1], b, e1);
                    }
                    // BN as packed fma (v_pk_fma_f32); pixels outside the image must come out 0 (the depthwise pads the EXPANDED map)
                    f32x2 t[4];
                    t[0] = (f32x2){e0[0], e0[1]} * (f32x2){se0[0], se0[1]} + (f32x2){he0[0], he0[1]};
                    t[1] = (f32x2){e0[2], e0[3]} * (f32x2){se0[2], se0[3]} + (f32x2){he0[2], he0[3]};
                    t[2] = (f32x2){e1[0], e1[1]} * (f32x2){se1[0], se1[1]} + (f32x2){he1[0], he1[1]};
                    t[3] = (f32x2){e1[2], e1[3]} * (f32x2){se1[2], se1[3]} + (f32x2){he1[2], he1[3]};
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[2 * e] = t[e][0];
                        v[2 * e + 1] = t[e][1];
                    }
                    const bool ok = (vm >> m) & 1u;
                    u32x4 o;
                    if constexpr (DT == PCV_F16 && ACT == PCV_ACT_RELU6) {
                        // fp16 + ReLU6: round, then clamp the packed pairs (pack2_clamp_f16): 8 instead of 20 instructions per pixel
                        // block; the outside-the-image mask is the upper bound again (6 or 0)
                        const uint32_t up = ok ? 0x46004600u : 0u;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = pack2_clamp_f16(v[2 * e], v[2 * e + 1], up);
                    } else if constexpr (ACT == PCV_ACT_RELU || ACT == PCV_ACT_RELU6) {
                        // the clamp's upper bound doubles as the mask: min(max(v, 0), ok ? 6 : 0) - one select per pixel block
                        const float hi = ok ? (ACT == PCV_ACT_RELU6 ? 6.f : INFINITY) : 0.f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = __builtin_elementwise_minimum(__builtin_elementwise_maximum(v[e], 0.f), hi);
                        if constexpr (ACT != PCV_ACT_RELU6) g1.see(v);             // (ReLU6: bounded by construction)
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = pack2<DT>(v[2 * e], v[2 * e + 1]);
                    } else {
                        apply_act8(v, act_e);
                        g1.see(v);
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = ok ? pack2<DT>(v[2 * e], v[2 * e + 1]) : 0u;
                    }
                    *reinterpret_cast<u32x4*>(const_cast<char*>(e_wr) + 16 * m * PITCH) = o;
                }
                // Schedule: the MFMAs of pixel block m + 1 go out with the BN / clamp / pack VALU work of block m (left alone the
                // compiler issues all 2 KA NPT MFMAs first and the VALU work after them: the two pipes of the SIMD never overlap
                // inside a wave, and with two waves per SIMD they rarely do across waves).
                if constexpr (ACT == PCV_ACT_RELU || ACT == PCV_ACT_RELU6) {
#pragma unroll
                    for (int m = 0; m < NPT; ++m) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 2 * KA, 0);
                        if (m > 0) {
                            __builtin_amdgcn_sched_group_barrier(0x002, S1VALU, 0);
                            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                        }
                    }
                    __builtin_amdgcn_sched_group_barrier(0x002, S1VALU, 0);
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- S2: depthwise 3x3 of the E chunk -> D chunk -------------------------------------------------------------------------
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                frag af[5];
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const uint32_t w16 = *reinterpret_cast<const uint16_t*>(a_w + (2 * j * CmidP + 32 * c + 16 * g) * 2);
                    const uint32_t val = w16 << a_sh;
                    u32x4 a4;
#pragma unroll
                    for (int i = 0; i < 4; ++i) a4[i] = val & am[i];
                    af[j] = __builtin_bit_cast(frag, a4);
                }
                const f32x4 sd = *reinterpret_cast<const f32x4*>(BNs + 2 * CmidP + 32 * c + 16 * g + 4 * fq);
                const f32x4 hd = *reinterpret_cast<const f32x4*>(BNs + 3 * CmidP + 32 * c + 16 * g + 4 * fq);
#pragma unroll
                for (int u = 0; u < R; ++u) {
                    f32x4 da = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < 5; ++j) {
                        uint32_t ba;
                        if constexpr (BTAB) {
                            ba = boff[u][j];
                        } else {
                            ba = boff[0][j] + (uint32_t)(u * S * IWS * PITCH);
                        }
                        const frag b = *reinterpret_cast<const frag*>(Es + (g == 0 ? ba : (SWZ ? (ba ^ 32u) : ba + 32u)));
                        da = Mma<DT>::run(af[j], b, da);
                    }
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = da[e] * sd[e] + hd[e];
                    u32x2 o;
                    if constexpr (DT == PCV_F16 && ACT == PCV_ACT_RELU6) {
                        o[0] = pack2_clamp_f16(v[0], v[1], 0x46004600u);
                        o[1] = pack2_clamp_f16(v[2], v[3], 0x46004600u);
                    } else {
                        mbw_act<ACT, 4>(v, act_d);
                        if constexpr (ACT != PCV_ACT_RELU6) g2.see(v);
                        o[0] = pack2<DT>(v[0], v[1]);
                        o[1] = pack2<DT>(v[2], v[3]);
                    }
                    *reinterpret_cast<u32x2*>((g == 0 ? d_wr0 : d_wr1) + (16 * u) * PITCH) = o;
                }
            }
            if constexpr (ACT != PCV_ACT_RELU6) {
                g1.commit(p.ovf);
                g2.commit(p.ovf);
            }
            // ---- S3: project GEMM, K step = this chunk ---------------------------------------------------------------------------------
            {
                frag wp[NRT];
#pragma unroll
                for (int i = 0; i < NRT; ++i) {
                    if constexpr (WLDS) wp[i] = *reinterpret_cast<const frag*>(wp_rd + ((c * NRT + i) * 16) * PITCH);
                    else wp[i] = __builtin_bit_cast(frag, wpn[i]);
                }
#pragma unroll
                for (int u = 0; u < R; ++u) {
                    const frag b = *reinterpret_cast<const frag*>(d_rd + (16 * u) * PITCH);
#pragma unroll
                    for (int i = 0; i < NRT; ++i) acc[i][u] = Mma<DT>::run(wp[i], b, acc[i][u]);
                }
            }
        }

        // ---- epilogue: BN (+ residual), 16-byte NHWC stores ---------------------------------------------------------------------------
        F16Guard<DT, true> guard;
#pragma unroll
        for (int ipp = 0; ipp < NRT / 2; ++ipp) {
            const int ch = 32 * ipp + 8 * fq;
            f32x4 sp[2], hp[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                sp[h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(sprsrc, (uint32_t)((ch + 4 * h) * 4), 0, 0));
                hp[h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(hprsrc, (uint32_t)((ch + 4 * h) * 4), 0, 0));
            }
#pragma unroll
            for (int u = 0; u < R; ++u) {
                const int ho = ho0 + u + R * pr, wo = wo0 + pc;
                const bool ok = ch < p.Cout && ho < p.Ho && wo < p.Wo;
                const uint32_t off = ok ? (uint32_t)(((((long)n * p.Ho + ho) * p.Wo + wo) * p.Cout + ch) * 2) : 0x80000000u;
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[2 * ipp][u][e] * sp[0][e] + hp[0][e];
                    v[4 + e] = acc[2 * ipp + 1][u][e] * sp[1][e] + hp[1][e];
                }
                apply_act8(v, act_p);
                if (p.res != nullptr) {
                    const u32x4 rv = __builtin_amdgcn_raw_buffer_load_b128(rrsrc, off, 0, 0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float r0, r1;
                        unpack2<DT>(rv[e], r0, r1);
                        v[2 * e] += r0;
                        v[2 * e + 1] += r1;
                    }
                    apply_act8(v, post);
                }
                guard.see(v);
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = pack2<DT>(v[2 * e], v[2 * e + 1]);
                __builtin_amdgcn_raw_buffer_store_b128(o, yrsrc, off, 0, 0);
            }
        }
        guard.commit(p.ovf);

        tile = ntile;
        if constexpr (XPRE) {
#pragma unroll
            for (int m = 0; m < NPT; ++m) xa[0][m] = xb[0][m];
            vm = vmn;
        }
    }
#endif  // __HIP_DEVICE_COMPILE__
}

// mbr.hpp - the fused inverted-residual unit (1x1 expand + BN + act -> depthwise 3x3 + BN + act -> 1x1 project + BN (+ skip);
// reference LinearBottleneck.forward, mobilenetv2.py:62-71, dwconv_block, common/conv.py:437-473) with the expanded tensor
// held in REGISTERS: no LDS traffic inside a tile.
//
// mbw.hpp's wave keeps its E (expanded) and D (depthwise output) tiles in LDS: per 32-channel chunk a wave writes 7 KB and reads
// 54 KB, S2 waits for S1's writes, S3 for S2's, and the 8 waves of a CU need ~250 B/clk of LDS at the rate the matrix pipe could
// run - round 4 measured the class at 0.13 of HBM with the matrix pipe 27 % busy and no unit saturated. Here:
//
//   tile   RO output rows x 14 output columns of one image per wave; its input window is (RO + 2) rows x 16 columns = ONE MFMA
//          pixel block per window row (lane l: window column l % 16, K quarter l / 16). Every stride-1 map of MobileNetV2 is a
//          multiple of 14 wide (112, 56, 28, 14).
//   x      global -> registers as B fragments (16 B = 8 channels of the lane's pixel), one per window row, live for all chunks; the
//          next tile's rows are requested in the last chunk, each behind the S1 that used the register last.
//   S1     E row = act(BN(W_exp[chunk] . x row)): two MFMAs; the packed result of lane (pixel, q) is channels 8 q .. 8 q + 7 of that
//          pixel (the packed weight rows are in MFMA order, igemm_conv.hpp). Pixels outside the image come out 0 (the clamp's
//          upper bound is the mask, as in mbw.hpp).
//   taps   the left / right neighbours of a pixel are the neighbouring LANES of its 16-lane row: two DPP moves per dword
//          (row_shr:1 / row_shl:1, zero fill = the halo lanes 0 and 15, whose outputs are never stored). The rows above / below
//          are other registers. So the B operand of the depthwise MFMA is assembled from registers:
//   S2     depthwise 3x3 as block-diagonal MFMAs over 16 channels (half g of the chunk: channels 8 q + 4 g + e) - on the SPARSE
//          matrix instruction v_smfmac_f32_16x16x64 (2:4 structured-sparse A at twice the K of the dense form for the same 16
//          cycles; operand layout probed by tests/tools/micro/smfmac_probe.cpp). A diagonal weight matrix has ONE non-zero per
//          group of four K values (K = tap x 4 channels of a lane), so it is exactly representable. One K = 64 step = 4 tap
//          slots x 16 channels: R(r) = {left, centre, right, 0} of window row r, and output row u = R(u) W0 + R(u+1) W1 + R(u+2) W2:
//          3 instructions (the dense form needs 5 for its 10 tap slots), R(r) shared by three output rows. The compressed
//          fragments are prebuilt once per block in LDS (6 KB per chunk) and read once per chunk.
//   S3     the packed S2 results of both halves ARE the B fragment of the project GEMM in natural K order (slot s of lane q =
//          channel 8 q + s): no LDS round trip. acc[Cout][RO x 16] += W_proj[:, chunk] . D
//   then BN (+ residual) and 16-byte NHWC stores of lanes 1 .. 14.
// Same rounding points as the three launches (E, D, y rounded to the storage type; fp32 accumulation everywhere).
// LDS holds only what every wave reads: the 1x1 weights (64-byte swizzled rows as in mbw.hpp), the diagonal fragments, BN constants.
#pragma once
#include <type_traits>
#include "pcv_common.hpp"
#include "igemm_conv.hpp"     // Mma<DT>
#include "mbconv.hpp"         // MbParams
#include "mbw.hpp"            // mbw_swz, mbw_act

struct MbrLds {
    int wexp, wproj, af, bn, bnp, xs, total;
};
// ka: K steps of the expand GEMM (Cin <= 32 ka); afl: the compressed depthwise fragments live in LDS too (else every wave reads its
// three fragments per half-chunk from the packed table in L2: units whose 6 KB per chunk do not fit)
// wel: the expand weights live in LDS (else every wave reads its fragments from L2 per half-chunk, one pass ahead: the 96 -> 576 -> 96
// units, whose two 1x1 matrices are 216 KB)
// xrows: window rows of a wave tile when x is staged through LDS (two buffers per wave, XL below), else 0
static inline __host__ __device__ MbrLds mbr_lds_layout(int nrt, int nChunks, int ka, bool afl, bool wel = true, int xrows = 0, int waves = 8) {
    MbrLds L;
    int o = 0;
    L.wexp = o; o += wel ? nChunks * ka * 32 * 64 : 0;  // [chunk][K step][32 rows][64 B]
    L.wproj = o; o += nChunks * nrt * 16 * 64;          // [chunk][nrt * 16 rows][64 B]
    L.af = o; o += afl ? nChunks * 6 * 1024 : 0;        // [chunk][half g][filter row][lane] 16 B: compressed diagonal depthwise fragments
    L.bn = o; o += nChunks * 4 * 32 * 4;                // [chunk][scale_e, shift_e, scale_d, shift_d][32 ch] fp32
    L.bnp = o; o += 2 * nrt * 16 * 4;                   // scale_p, shift_p
    L.xs = o; o += waves * 2 * xrows * 1024;            // [wave][buffer][window row][lane] 16 B
    L.total = o;
    return L;
}
// timing experiments only (tests/tools/sh/mbr_variants.sh builds one library per value; results are WRONG with any bit set):
// 1: one S2 MFMA instead of five   2: no DPP moves   4: no epilogue   8: no S3   16: no S1 epilogue (BN / clamp)   32: no x loads in the loop
#ifndef MBR_DBG
#define MBR_DBG 0
#endif

constexpr int kMbrCols = 14;                            // output columns of a wave tile (window = 16 columns = one MFMA pixel block)
// Stride 2 with two projection tiles (Cout <= 32: MobileNetV2's 16 -> 96 -> 24 and 24 -> 144 -> 32, 29 % of its fused-unit time): the window
// row is split by column PARITY - an even-column pixel block and an odd-column pixel block (csrc/gconv3x3r.hpp does the same) - so that the
// {left, centre, right} tuple of output column j is {odd lane j, even lane j, odd lane j + 1}: 15 outputs per pixel block instead of 7
// (the depthwise and projection MFMAs and the D epilogue per output halve, one DPP move instead of two; the expand work per output is
// the same 32 / 15 against 16 / 7 input columns). Its x fragments double (72 registers), which the four-tile instances do not have.
constexpr bool mbr_parity_split(int S, int NRT) { return S == 2 && NRT == 2; }
constexpr int mbr_out_cols(int S, int NRT) { return S == 1 ? kMbrCols : (mbr_parity_split(S, NRT) ? 15 : kMbrCols / 2); }

// v_smfmac_f32_16x16x64_{f16,bf16}: D(16x16) += A(16x64, 2:4 sparse) . B(64x16). Operand layout as measured on gfx950
// (tests/tools/micro/smfmac_probe.cpp): K = two halves of 32. B lane (column n = l % 16, kq = l / 16) element e (16 per lane): half
// e / 8, dense K = 8 kq + e % 8 inside the half - i.e. FOUR groups of four consecutive K per lane. A lane (row i = l % 16, q = l / 16)
// holds 8 compressed values = 4 slot pairs m: pair m keeps two of the four dense values of B's group (kq, gb) with
// kq = 2 (q & 1) + m / 2, gb = 2 (q / 2) + m % 2; every 4-bit field p0 | p1 << 2 of the index register names their positions
// (slot 2 m <- position p0, slot 2 m + 1 <- position p1). 16.2 cycles per instruction, as the dense 16x16x32.
template <int DT> struct MmaSp;
template <> struct MmaSp<PCV_F16> {
    typedef __attribute__((ext_vector_type(16))) _Float16 bfrag;
    static __device__ __forceinline__ f32x4 run(const f16x8& a, const bfrag& b, f32x4 c, int idx) {
        return __builtin_amdgcn_smfmac_f32_16x16x64_f16(a, b, c, idx, 0, 0);
    }
};
template <> struct MmaSp<PCV_BF16> {
    typedef __attribute__((ext_vector_type(8))) __bf16 afrag;
    typedef __attribute__((ext_vector_type(16))) __bf16 bfrag;
    static __device__ __forceinline__ f32x4 run(const s16x8& a, const bfrag& b, f32x4 c, int idx) {
        return __builtin_amdgcn_smfmac_f32_16x16x64_bf16(__builtin_bit_cast(afrag, a), b, c, idx, 0, 0);
    }
};
typedef __attribute__((ext_vector_type(8))) uint32_t u32x8;

// fp16 pair {clamp(a0 s0 + h0, 0, 1), clamp(a1 s1 + h1, 0, 1)}: hipcc selects v_fma_mixlo_f16 / v_fma_mixhi_f16 with the clamp
// modifier for exactly this shape (fp32 FMA, rounded once, the PACKED pair clamped: 0 and 1 are fp16 values, so clamping before or
// after the rounding is the same)
__device__ __forceinline__ uint32_t mbr_bn_clamp01(float a0, float s0, float h0, float a1, float s1, float h1) {
    typedef __attribute__((ext_vector_type(2))) _Float16 h2;
    h2 r = {(_Float16)__builtin_fmaf(a0, s0, h0), (_Float16)__builtin_fmaf(a1, s1, h1)};
    const h2 zero = {(_Float16)0.f, (_Float16)0.f}, one = {(_Float16)1.f, (_Float16)1.f};
    r = __builtin_elementwise_min(__builtin_elementwise_max(r, zero), one);
    return __builtin_bit_cast(uint32_t, r);
}

template <int CTRL> __device__ __forceinline__ uint32_t mbr_dpp(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);      // bound_ctrl: lanes shifted in read 0
}

// NRT: 16-row tiles of the project GEMM (2: Cout <= 32, 4: <= 64, 6: <= 96); ACT: activation of the expand and depthwise stages when
// both are ReLU or both ReLU6, -1 = read p.act_e / p.act_d; RO: output rows of a wave tile; S: stride; KA: K steps of the expand GEMM
// (Cin <= 32 KA); AFL: depthwise fragments in LDS (else read from the packed table per half-chunk); WAVES: waves per block (8: two per
// SIMD at 256 registers; 4: one per SIMD with the 512-register budget, for the units whose accumulators + x fragments need it);
// WEL: expand weights in LDS; XL: the x window goes global -> LDS by LDS-DMA, a whole tile ahead, into one of two wave-private buffers
// (units with so few chunks that the register form's prefetch - issued inside the last chunk - arrives late: 32 -> 32 -> 16 at 112x112
// is ONE chunk per tile and waited ~2.5 K of its 10.8 K cycles for x). blockDim.x = 64 WAVES.
template <int DT, int NRT, int ACT, int RO, int S = 1, int KA = 1, bool AFL = true, int WAVES = 8, bool WEL = true, bool XL = false>
__global__ __launch_bounds__(64 * WAVES) void mbr_kernel(const MbParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    // Stride 2 keeps the window layout (16 consecutive input columns per lane row, so the taps of the output centred on lane l are
    // still lanes l - 1, l, l + 1): the depthwise stage runs on every lane, the outputs are the 7 odd lanes 1 .. 13 (input columns
    // w0 + 0, 2 .. 12) and every second window row; window row r is filter row 0 of output r / 2 and 2 of r / 2 - 1 (r even) or 1 of
    // (r - 1) / 2 (r odd). 2 RO + 1 window rows, no skip tensor.
    constexpr int NR = S == 1 ? RO + 2 : 2 * RO + 1;    // window rows
    constexpr bool PS = mbr_parity_split(S, NRT);        // window rows as an even-column and an odd-column pixel block
    constexpr int NBK = PS ? 2 : 1;                      // pixel blocks per window row
    constexpr int OC = mbr_out_cols(S, NRT);             // output columns of a tile
    // fp16 + ReLU6 (MobileNetV2's default mode): E and D are kept as E / 6 and D / 6 in [0, 1], so that BN + rounding + clamp of a pair
    // is TWO instructions, v_fma_mixlo_f16 / v_fma_mixhi_f16 with the clamp modifier (the generic path: two FMAs, a conversion, two
    // packed clamps - and a dependent chain of four where the loop is bound by exactly such chains: removing the 72 DPP moves of a
    // chunk changed nothing, removing the S1 epilogue took 18 % off). The factors are folded into the fp32 BN constants when they are
    // copied to LDS: scale_e / 6, shift_e / 6, shift_d / 6, 6 scale_p. Rounding E / 6 instead of E is a different but equally good
    // 11-bit rounding of the same fp32 value (the tests bound the difference to the three separate launches at 2 ulp of the output).
    constexpr bool FAST = DT == PCV_F16 && ACT == PCV_ACT_RELU6;
    if constexpr (FAST) __builtin_amdgcn_s_setreg(1 | (8 << 6), 0);      // MODE.DX10_CLAMP = 0: the clamp modifier passes NaN through (as torch's hardtanh)
    typedef typename Mma<DT>::frag frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(!XL || KA == 1, "x through LDS: one K step");
    const MbrLds L = mbr_lds_layout(NRT, p.nChunks, KA, AFL, WEL, XL ? NR : 0, WAVES);
    // fragments that come from L2 are requested one half-chunk pass ahead into the other of two register sets (only where the registers
    // exist: the one-wave-per-SIMD form)
    constexpr bool PF = WAVES == 4 && (!WEL || !AFL);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nWaves = blockDim.x >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    char* const Wes = smem + L.wexp;
    char* const Wps = smem + L.wproj;
    char* const Afs = smem + L.af;
    float* const BNs = reinterpret_cast<float*>(smem + L.bn);
#ifdef MBR_CYCLES      // diagnostic build (tests/tools/mbr_cycles.py): shader-cycle stamps of the prologue and of this wave's second tile
    const uint64_t k0__ = __builtin_amdgcn_s_memtime();
    uint64_t k1__ = 0;
    int ntile__ = 0;
#endif

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res), 0, p.res ? p.y_bytes : 0, 0x00020000);

    // ---- the unit's weights and BN constants -> LDS, once per block -------------------------------------------------------------------
    // ONE loop over every 16-byte piece of the five tables, four independent loads in flight per thread: as five loops of
    // load -> store the prologue measured 10 K cycles (every iteration a full L2 latency), 4 - 14 % of a launch. Pieces beyond a
    // table's end (padded rows / channels) are stored as zeros. The depthwise fragments come ready-made from the packed blob
    // (pack_dw_sparse_kernel, aux_kernels.hpp).
    {
        const int nA = WEL ? p.nChunks * KA * 128 : 0, nB = p.nChunks * NRT * 64, nC = AFL ? p.nChunks * 6 * 64 : 0, nD = p.nChunks * 32, nE = 8 * NRT;
        const int total = nA + nB + nC + nD + nE;
        for (int base = tid; base < total; base += 4 * blockDim.x) {
            u32x4 v[4];
            float fold[4];
            char* dst[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int i = base + k * blockDim.x;
                const char* src = nullptr;
                bool ok = false;
                dst[k] = nullptr;
                if (i < nA) {                                                      // expand weights: rows 32 c + r in MFMA order, K step ks
                    const int slot = i & 3, r = (i >> 2) & 31, ks = (i >> 7) % KA, c = (i >> 7) / KA;
                    const uint32_t off = (uint32_t)(((32 * c + r) * p.Kpad1 + 32 * ks + 8 * slot) * 2);
                    src = static_cast<const char*>(p.w_exp) + off; ok = off + 16u <= p.wexp_bytes;
                    dst[k] = Wes + ((c * KA + ks) * 32 + r) * 64 + ((slot ^ mbw_swz<true>(r)) << 4);
                } else if ((i -= nA) < nB) {                                       // projection weights: K slice 32 c of every row
                    const int slot = i & 3, row = (i >> 2) % (NRT * 16), c = (i >> 2) / (NRT * 16);
                    const uint32_t off = (uint32_t)((row * p.Kpad2 + 32 * c + 8 * slot) * 2);
                    src = static_cast<const char*>(p.w_proj) + off; ok = off + 16u <= p.wproj_bytes;
                    dst[k] = Wps + (c * NRT * 16 + row) * 64 + ((slot ^ mbw_swz<true>(row)) << 4);
                } else if ((i -= nB) < nC) {                                       // compressed depthwise fragments, as packed
                    src = static_cast<const char*>(p.w_dwsp) + i * 16; ok = true;
                    dst[k] = Afs + i * 16;
                } else if ((i -= nC) < nD) {                                       // BN of the expand / depthwise stages -> [chunk][which][32]
                    const int pc = i & 7, which = (i >> 3) & 3, ch = 32 * (i >> 5) + 4 * pc;
                    const float* arr = which == 0 ? p.scale_e : which == 1 ? p.shift_e : which == 2 ? p.scale_d : p.shift_d;
                    src = reinterpret_cast<const char*>(arr + ch); ok = arr != nullptr && ch < p.Cmid;
                    dst[k] = reinterpret_cast<char*>(BNs) + i * 16;
                } else if ((i -= nD) < nE) {                                       // BN of the projection
                    const int which = i / (NRT * 4), ch = (i % (NRT * 4)) * 4;
                    src = reinterpret_cast<const char*>((which ? p.shift_p : p.scale_p) + ch); ok = ch < p.Cout;
                    dst[k] = smem + L.bnp + i * 16;
                }
                v[k] = ok ? *reinterpret_cast<const u32x4*>(src) : (u32x4){0u, 0u, 0u, 0u};
                fold[k] = 1.f;
                if constexpr (FAST) {
                    const int j = base + k * (int)blockDim.x - nA - nB - nC;      // position inside the BN tables
                    if (j >= 0 && j < nD) fold[k] = ((j >> 3) & 3) == 2 ? 1.f : (1.f / 6.f);
                    else if (j >= nD && j < nD + nE) fold[k] = (j - nD) < NRT * 4 ? 6.f : 1.f;
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if constexpr (FAST) {
                    if (fold[k] != 1.f) v[k] = __builtin_bit_cast(u32x4, __builtin_bit_cast(f32x4, v[k]) * fold[k]);
                }
                if (dst[k] != nullptr) *reinterpret_cast<u32x4*>(dst[k]) = v[k];
            }
        }
    }
    __syncthreads();                                                               // the only barrier of the kernel

#ifdef MBR_CYCLES
    k1__ = __builtin_amdgcn_s_memtime();
#endif
    const ActClamp act_e = make_act(p.act_e), act_d = make_act(p.act_d), act_p = make_act(p.act_p), post = make_act(p.post);

    const int fsw = (fq ^ mbw_swz<true>(fr)) << 4;                                   // fragment access of row (16 k + fr), slot fq
    const char* const we_rd = Wes + fr * 64 + fsw;
    const char* const wp_rd = Wps + fr * 64 + fsw;
    const char* const af_rd = Afs + lane * 16;
    const __amdgpu_buffer_rsrc_t wersrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w_exp), 0, p.wexp_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t afrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w_dwsp), 0, (uint32_t)(p.nChunks * 6 * 1024), 0x00020000);
    // sparse index of this lane's A rows (MmaSp): every group keeps positions (i % 4, 3) - the weight in the pair's first slot, a zero in
    // the second - or (0, 3) with the weight second for i % 4 == 3; all eight 4-bit fields alike
    const int spidx = (((fr & 3) == 3 ? 0 : (fr & 3)) | (3 << 2)) * 0x11111111;

    const int nWavesAll = gridDim.x * nWaves;
    int tile = blockIdx.x * nWaves + wave;
    const bool resx = (MBR_DBG & 64) == 0 && S == 1 && NRT == 2 * KA && p.res != nullptr && p.res == p.x;                    // the skip tensor is the unit's input (Cout == Cin)

    // A tile's position, decoded ONCE (three scalar divisions; decoded inside every row load the epilogue's prefetch alone was ~700
    // scalar instructions per tile): image, first output row, this lane's window column and the byte offset of its pixel in window row 0
    struct TilePos { int n, h0, wi; bool colok, colok1; int off0; };        // (PS: wi / colok / off0 of the EVEN block; the odd block is one column to the left)
    auto decode = [&](int t) __attribute__((always_inline)) -> TilePos {
        TilePos T;
        const int tw = t % p.tilesW;
        const int t2 = t / p.tilesW;
        const int th = t2 % p.tilesH;
        T.n = t2 / p.tilesH;
        T.h0 = th * RO;                                                             // first OUTPUT row
        T.wi = PS ? tw * OC * 2 + 2 * fr : tw * OC * S - 1 + fr;                   // this lane's INPUT column
        T.colok = (t < p.nTiles) & ((unsigned)T.wi < (unsigned)p.W);
        T.colok1 = PS && (t < p.nTiles) & ((unsigned)(T.wi - 1) < (unsigned)p.W);
        T.off0 = (((T.n * p.H + T.h0 * S - 1) * p.W + T.wi) * p.Cin + 8 * fq) * 2;                          // < 2 GiB: checked by the host
        return T;
    };
    const int rowpitch = p.W * p.Cin * 2;
    // x fragment of window row r (window rows h0 - 1 .. h0 + RO, columns w0 - 1 .. w0 + 14)
    auto load_x = [&](const TilePos& T, int r, int b, int ks) __attribute__((always_inline)) -> u32x4 {        // (b: 0 = the block / the even block, 1 = the odd block)
        const bool ok = (b ? T.colok1 : T.colok) & ((unsigned)(T.h0 * S - 1 + r) < (unsigned)p.H) & (32 * ks + 8 * fq < p.Cin);
        return __builtin_amdgcn_raw_buffer_load_b128(xrsrc, ok ? (uint32_t)(T.off0 - b * p.Cin * 2 + r * rowpitch + 64 * ks) : 0x80000000u, 0, 0);
    };

    // fragments of half-chunk pass h = 2 c + g that do not live in LDS
    auto fetch_we = [&](int h, frag (&w)[KA]) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < KA; ++ks)
            w[ks] = __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(wersrc, (uint32_t)(((16 * h + fr) * p.Kpad1 + 32 * ks + 8 * fq) * 2), 0, 0));
    };
    auto fetch_af = [&](int h, frag (&a)[3]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 3; ++j) a[j] = __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(afrsrc, (uint32_t)(((h * 3 + j) * 64 + lane) * 16), 0, 0));
    };
    frag weG[PF ? 2 : 1][KA], afG[PF ? 2 : 1][3];
    if constexpr (PF) {
        if constexpr (!WEL) fetch_we(0, weG[0]);
        if constexpr (!AFL) fetch_af(0, afG[0]);
    }

    // the B tuples of the sparse MFMAs, {left, centre, right, 0} x 2 dwords: slots 0 .. 5 are rewritten per window row, the padding slots
    // 6, 7 stay zero (as fresh values per row they cost two v_mov per tuple plus the compiler's tuple copies: 62 of a chunk's 320 VALU)
    u32x8 RT[2] = {(u32x8){0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}, (u32x8){0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}};
    TilePos cur = decode(tile);
    u32x4 xr[XL ? 1 : NR][NBK][KA];
    // XL: this wave's two x buffers, [row][lane] 16 B each; a row is one LDS-DMA piece (out-of-range lanes are written as zeros)
    typedef __attribute__((address_space(3))) char lds_char;
    char* const xs = smem + L.xs + wave * 2 * NR * 1024;
    auto stage_x = [&](const TilePos& T, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const bool ok = T.colok & ((unsigned)(T.h0 * S - 1 + r) < (unsigned)p.H) & (8 * fq < p.Cin);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_char*)(size_t)PCV_LDS(xs + (buf * NR + r) * 1024), 16,
                                                     ok ? (uint32_t)(T.off0 + r * rowpitch) : 0x80000000u, 0, 0, 0);
        }
    };
    int xbuf = 0;
    if (tile < p.nTiles) {
        if constexpr (XL) {
            stage_x(cur, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // (the first tile has no stores in front of it to count)
        } else {
#pragma unroll
            for (int r = 0; r < NR; ++r)
#pragma unroll
                for (int b = 0; b < NBK; ++b)
#pragma unroll
                    for (int ks = 0; ks < KA; ++ks) xr[r][b][ks] = load_x(cur, r, b, ks);
        }
    }

    while (tile < p.nTiles) {
        const int ntile = tile + nWavesAll;
        const TilePos nxt = decode(ntile);
        const int n = cur.n, h0 = cur.h0, wi = cur.wi;
        const bool colok = (unsigned)wi < (unsigned)p.W;
        const bool colok1 = PS && (unsigned)(wi - 1) < (unsigned)p.W;
        const uint32_t cmask[2] = {colok ? 0xFFFFFFFFu : 0u, colok1 ? 0xFFFFFFFFu : 0u};
        if constexpr (XL) {
            // this tile's window has landed: behind its pieces (issued a tile ago) only the previous tile's RO * NRT / 2 stores went out
            // (vmcnt retires in order; a skip tensor that is not x is loaded AND consumed in front of them). Then the next tile's pieces
            // go into the other buffer, whose last reader was the previous tile's S1.
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RO * NRT / 2) : "memory");
            stage_x(nxt, xbuf ^ 1);
        }
        const char* const xrd = xs + xbuf * NR * 1024 + lane * 16;

#ifdef MBR_CYCLES
        const bool stamp__ = p.dbg != nullptr && ntile__ == 1;
        ++ntile__;
        uint64_t c0__ = 0, c1__ = 0;
        if (stamp__) c0__ = __builtin_amdgcn_s_memtime();
#endif
        f32x4 acc[NRT][RO];
#pragma unroll
        for (int i = 0; i < NRT; ++i)
#pragma unroll
            for (int u = 0; u < RO; ++u) acc[i][u] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
        for (int c = 0; c < p.nChunks; ++c) {
            const bool last = c + 1 == p.nChunks;
            F16Guard<DT, true> g1, g2;                                // fp16 range checks of this chunk's E / D values (unbounded activations only)
            u32x2 d0[RO];                                             // half 0 of the D rows, kept until half 1 completes the K step of S3
            // The two 16-channel halves of the chunk are two passes over the tile: S1's two MFMAs are independent (fragment g of the
            // expand weights gives exactly the channels 8 q + 4 g + e that S2's half g convolves), so a pass holds 3 diagonal
            // fragments, one expand fragment and one half's BN constants (both halves at once spilled: 256 registers).
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                frag af[3], we[KA];
                if constexpr (PF) {                                     // this pass from set g, the next pass (the next tile's first behind the last) into set g ^ 1
                    const int hn = (2 * c + g + 1 == 2 * p.nChunks) ? 0 : 2 * c + g + 1;
                    if constexpr (!WEL) fetch_we(hn, weG[g ^ 1]);
                    if constexpr (!AFL) fetch_af(hn, afG[g ^ 1]);
                }
#pragma unroll
                for (int ks = 0; ks < KA; ++ks) {
                    if constexpr (WEL) we[ks] = *reinterpret_cast<const frag*>(we_rd + ((c * KA + ks) * 32 + 16 * g) * 64);
                    else if constexpr (PF) we[ks] = weG[g][ks];
                }
                if constexpr (!WEL && !PF) fetch_we(2 * c + g, we);
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    if constexpr (AFL) af[j] = *reinterpret_cast<const frag*>(af_rd + ((c * 2 + g) * 3 + j) * 1024);
                    else if constexpr (PF) af[j] = afG[g][j];
                }
                if constexpr (!AFL && !PF) fetch_af(2 * c + g, af);
                // BN constants of this lane's 4 channels 32 c + 8 fq + 4 g + e: the same channels in S1 (accumulator rows) and S2
                f32x4 se = *reinterpret_cast<const f32x4*>(BNs + 128 * c + 8 * fq + 4 * g);
                f32x4 he = *reinterpret_cast<const f32x4*>(BNs + 128 * c + 32 + 8 * fq + 4 * g);
                if constexpr (FAST && !PS) {                           // columns outside the image: E = clamp(0 . acc + 0) = 0 (PS: two blocks, masked below)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        se[e] = colok ? se[e] : 0.f;
                        he[e] = colok ? he[e] : 0.f;
                    }
                }
                const f32x4 sd = *reinterpret_cast<const f32x4*>(BNs + 128 * c + 64 + 8 * fq + 4 * g);
                const f32x4 hd = *reinterpret_cast<const f32x4*>(BNs + 128 * c + 96 + 8 * fq + 4 * g);
                frag wp[NRT];
                if (g == 1) {
#pragma unroll
                    for (int i = 0; i < NRT; ++i) wp[i] = *reinterpret_cast<const frag*>(wp_rd + ((c * NRT + i) * 16) * 64);
                }

                // Window row r is used the moment it exists: its tuple R(r) = {left, centre, right, 0} is filter row 2 of output row r - 2,
                // filter row 1 of output row r - 1 and filter row 0 of output row r - three sparse MFMAs on three DIFFERENT accumulators
                // (as a chain of three on one accumulator, written when the last of three window rows arrived, every removed MFMA
                // saved twice its pipe time: the chain was the critical path; in-kernel stamps).
                f32x4 da[RO + 2];                                             // (RO used)
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    // ---- S1: window row r of the E half-chunk ----------------------------------------------------------------------
                    u32x8 fresh = (u32x8){0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
                    u32x8& b8 = KA == 1 ? RT[r & 1] : fresh;             // (two persistent tuples, alternating; with two expand K steps their 16 registers spill)
                    {
                        uint32_t ob[NBK][2];
#pragma unroll
                        for (int bk = 0; bk < NBK; ++bk) {
                        f32x4 e0 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int ks = 0; ks < KA; ++ks) {
                            if constexpr (XL) e0 = Mma<DT>::run(we[ks], *reinterpret_cast<const frag*>(xrd + r * 1024), e0);
                            else e0 = Mma<DT>::run(we[ks], __builtin_bit_cast(frag, xr[r][bk][ks]), e0);
                        }
                        // the next tile's row into the registers this S1 used last (rows that are the unit's skip tensor: behind the epilogue)
                        if (!XL && (MBR_DBG & 32) == 0 && g == 1 && last && (!resx || r == 0 || r == NR - 1)) {
#pragma unroll
                            for (int ks = 0; ks < KA; ++ks) xr[r][bk][ks] = load_x(nxt, r, bk, ks);
                        }
                        uint32_t* const o = ob[bk];
                        if constexpr (FAST) {
                            const uint32_t rowmask = (unsigned)(h0 * S - 1 + r) < (unsigned)p.H ? 0xFFFFFFFFu : 0u;      // (scalar) rows outside the image
                            const uint32_t m = PS ? (rowmask & cmask[bk]) : rowmask;
                            o[0] = mbr_bn_clamp01(e0[0], se[0], he[0], e0[1], se[1], he[1]) & m;
                            o[1] = mbr_bn_clamp01(e0[2], se[2], he[2], e0[3], se[3], he[3]) & m;
                        } else {
                            // (four v_fma_f32, not two v_pk_fma_f32: beside MFMAs a packed fp32 instruction costs several plain ones -
                            // MI355X_MICROARCH.md, "price of one filler"; mbr_*.hip are built with -fno-slp-vectorize for the same reason)
                            float v[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(e0[e], se[e], he[e]);
                            const bool ok = (bk ? colok1 : colok) & ((unsigned)(h0 * S - 1 + r) < (unsigned)p.H);   // E is zero outside the image (the depthwise pads the EXPANDED map)
                            if constexpr (ACT == PCV_ACT_RELU || ACT == PCV_ACT_RELU6) {
                                const float hi = ok ? (ACT == PCV_ACT_RELU6 ? 6.f : INFINITY) : 0.f;
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = __builtin_elementwise_minimum(__builtin_elementwise_maximum(v[e], 0.f), hi);
                                if constexpr (ACT != PCV_ACT_RELU6 && (MBR_DBG & 128) == 0) g1.see(v);
                                o[0] = pack2<DT>(v[0], v[1]);
                                o[1] = pack2<DT>(v[2], v[3]);
                            } else {
                                apply_actn<4>(v, act_e);
                                if constexpr ((MBR_DBG & 128) == 0) g1.see(v);
                                o[0] = ok ? pack2<DT>(v[0], v[1]) : 0u;
                                o[1] = ok ? pack2<DT>(v[2], v[3]) : 0u;
                            }
                        }
                        if constexpr ((MBR_DBG & 16) != 0) {
                            o[0] = __float_as_uint(e0[0]) ^ __float_as_uint(e0[1]);
                            o[1] = __float_as_uint(e0[2]) ^ __float_as_uint(e0[3]);
                        }
                        }
                        if constexpr (PS) {
                            // output column j = lane j: left tap = input column 2 j - 1 = odd lane j, centre = even lane j, right = odd lane j + 1
                            b8[0] = ob[1][0]; b8[1] = ob[1][1];
                            b8[2] = ob[0][0]; b8[3] = ob[0][1];
                            b8[4] = mbr_dpp<0x101>(ob[1][0]); b8[5] = mbr_dpp<0x101>(ob[1][1]);  // row_shl:1
                        } else if constexpr ((MBR_DBG & 2) != 0) {
                            b8[0] = ob[0][0]; b8[1] = ob[0][1]; b8[2] = ob[0][0]; b8[3] = ob[0][1]; b8[4] = ob[0][0]; b8[5] = ob[0][1];
                        } else {
                            b8[0] = mbr_dpp<0x111>(ob[0][0]); b8[1] = mbr_dpp<0x111>(ob[0][1]);  // row_shr:1 = the pixel to the left
                            b8[2] = ob[0][0]; b8[3] = ob[0][1];
                            b8[4] = mbr_dpp<0x101>(ob[0][0]); b8[5] = mbr_dpp<0x101>(ob[0][1]);  // row_shl:1 = the pixel to the right
                        }
                    }
                    // ---- S2: filter row 2 of output row r - 2 (complete behind it), 1 of r - 1, 0 of r ------------------------------------
                    const typename MmaSp<DT>::bfrag R = __builtin_bit_cast(typename MmaSp<DT>::bfrag, b8);
                    // (output row that window row r completes: -1 = none)
                    const int udone = S == 1 ? r - 2 : ((r & 1) == 0 ? r / 2 - 1 : -1);
                    if constexpr (S == 1) {
                        if (r >= 2) da[r - 2] = MmaSp<DT>::run(af[2], R, da[r - 2], spidx);
                        if ((MBR_DBG & 1) == 0 && r >= 1 && r - 1 < RO) da[r - 1] = MmaSp<DT>::run(af[1], R, da[r - 1], spidx);
                        if (r < RO) {
                            if constexpr ((MBR_DBG & 1) == 0) da[r] = MmaSp<DT>::run(af[0], R, (f32x4){0.f, 0.f, 0.f, 0.f}, spidx);
                            else da[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
                        }
                    } else if ((r & 1) == 0) {
                        if (r >= 2) da[r / 2 - 1] = MmaSp<DT>::run(af[2], R, da[r / 2 - 1], spidx);
                        if (r / 2 < RO) da[r / 2] = MmaSp<DT>::run(af[0], R, (f32x4){0.f, 0.f, 0.f, 0.f}, spidx);
                    } else {
                        da[r / 2] = MmaSp<DT>::run(af[1], R, da[r / 2], spidx);
                    }
                    // ---- BN + act of the finished row (+ S3 behind the second half) -----------------------------------------------------
                    if (udone >= 0) {
                        const int u = udone;
                        u32x2 od;
                        if constexpr (FAST) {
                            od[0] = mbr_bn_clamp01(da[u][0], sd[0], hd[0], da[u][1], sd[1], hd[1]);
                            od[1] = mbr_bn_clamp01(da[u][2], sd[2], hd[2], da[u][3], sd[3], hd[3]);
                        } else {
                            float v[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(da[u][e], sd[e], hd[e]);
                            mbw_act<ACT, 4>(v, act_d);
                            if constexpr (ACT != PCV_ACT_RELU6 && (MBR_DBG & 256) == 0) g2.see(v);
                            od[0] = pack2<DT>(v[0], v[1]);
                            od[1] = pack2<DT>(v[2], v[3]);
                        }
                        if (g == 0) {
                            d0[u] = od;
                        } else {
                            const frag b = __builtin_bit_cast(frag, (u32x4){d0[u][0], d0[u][1], od[0], od[1]});
#pragma unroll
                            for (int i = 0; i < ((MBR_DBG & 8) ? 1 : NRT); ++i) acc[i][u] = Mma<DT>::run(wp[i], b, acc[i][u]);
                        }
                    }
                }
            }
            if constexpr (ACT != PCV_ACT_RELU6) {
                g1.commit(p.ovf);
                g2.commit(p.ovf);
            }
        }

#ifdef MBR_CYCLES
        if (stamp__) c1__ = __builtin_amdgcn_s_memtime();
#endif
        // ---- epilogue: BN (+ residual), 16-byte NHWC stores of the 14 inner lanes ------------------------------------------------------
        // The skip tensor of a LinearBottleneck is the unit's input: window row u + 1 of x, which this lane still holds as a B
        // fragment (same pixel, same 8 channels). Loaded from memory row by row, behind a uniform branch each, the seven dependent
        // round trips were 28 % of a tile (in-kernel stamps); a skip tensor that is NOT x is requested for all rows up front.
        F16Guard<DT, true> guard;
        // plain: no activation behind the projection or behind the skip add (every LinearBottleneck): no uniform branches per row
        auto epilogue = [&](auto plain) __attribute__((always_inline)) {
        const bool lane_out = PS ? (fr < OC) : ((fr >= 1) & (fr <= kMbrCols) & colok & (S == 1 || (fr & 1)));
        const int wo = S == 1 ? wi : (wi >> 1);                                   // (stride 2: input column w0 + 2 j = lane 2 j + 1; parity split: even lane j)
        const float* const BNp = reinterpret_cast<const float*>(smem + L.bnp);
#pragma unroll
        for (int ipp = 0; ipp < ((MBR_DBG & 4) ? 0 : NRT / 2); ++ipp) {
            const int ch = 32 * ipp + 8 * fq;
            f32x4 sp[2], hp[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                sp[h] = *reinterpret_cast<const f32x4*>(BNp + ch + 4 * h);
                hp[h] = *reinterpret_cast<const f32x4*>(BNp + NRT * 16 + ch + 4 * h);
            }
#pragma unroll
            for (int u = 0; u < RO; ++u) {
                const int ho = h0 + u;
                const bool ok = lane_out & (ch < p.Cout) & (ho < p.Ho) & (wo < p.Wo);
                const uint32_t off = ok ? (uint32_t)(((((long)n * p.Ho + ho) * p.Wo + wo) * p.Cout + ch) * 2) : 0x80000000u;
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[2 * ipp][u][e] * sp[0][e] + hp[0][e];
                    v[4 + e] = acc[2 * ipp + 1][u][e] * sp[1][e] + hp[1][e];
                }
                if constexpr (!decltype(plain)::value) apply_act8(v, act_p);
                if (p.res != nullptr) {                                  // (x as the skip tensor: channels 32 ipp + 8 fq = K step ipp of window row u + 1)
                    u32x4 r4;
                    if constexpr (XL) r4 = resx ? *reinterpret_cast<const u32x4*>(xrd + (u + 1) * 1024) : __builtin_amdgcn_raw_buffer_load_b128(rrsrc, off, 0, 0);
                    else r4 = resx ? xr[S == 1 ? u + 1 : 0][0][ipp < KA ? ipp : 0] : __builtin_amdgcn_raw_buffer_load_b128(rrsrc, off, 0, 0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float r0, r1;
                        unpack2<DT>(r4[e], r0, r1);
                        v[2 * e] += r0;
                        v[2 * e + 1] += r1;
                    }
                    if constexpr (!decltype(plain)::value) apply_act8(v, post);
                }

                if constexpr ((MBR_DBG & 512) == 0) { if (ok) guard.see(v); }         // (halo lanes hold garbage that is never stored)
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = pack2<DT>(v[2 * e], v[2 * e + 1]);
                __builtin_amdgcn_raw_buffer_store_b128(o, yrsrc, off, 0, 0);
            }
        }
        };
        if (p.act_p == PCV_ACT_NONE && p.post == PCV_ACT_NONE) epilogue(std::true_type{});
        else epilogue(std::false_type{});
        // the rows of the next tile that were this tile's skip tensor: requested behind the epilogue, all at once. (Requested row by row
        // inside it, each behind the row that consumed it, one instantiation - fp16, launch-time activations - produced wrong,
        // run-to-run different output rows 0 - 2 of every tile although the loads only ever feed the NEXT tile; neither the matrix-pipe
        // distances nor the source explain it. tests: the "keep" shapes of test_mbconv_fused_matches_separate_launches_and_oracle.)
        if constexpr (S == 1 && !XL) {
            if ((MBR_DBG & 32) == 0 && resx) {
#pragma unroll
                for (int r = 1; r <= RO; ++r)
#pragma unroll
                    for (int ks = 0; ks < KA; ++ks) xr[r][0][ks] = load_x(nxt, r, 0, ks);
            }
        }
        guard.commit(p.ovf);
        if constexpr ((MBR_DBG & 4) != 0) {                             // keep the accumulators alive
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < NRT; ++i)
#pragma unroll
                for (int u = 0; u < RO; ++u) t += acc[i][u][0] + acc[i][u][1] + acc[i][u][2] + acc[i][u][3];
            if (t == 123.456f) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(t), yrsrc, 0, 0, 0);
        }
#ifdef MBR_CYCLES
        if (stamp__ && lane == 0) {
            const uint64_t c2 = __builtin_amdgcn_s_memtime();
            uint32_t* d = p.dbg + (blockIdx.x * 8 + wave) * 4;
            d[0] = (uint32_t)(k1__ - k0__); d[1] = (uint32_t)(c1__ - c0__); d[2] = (uint32_t)(c2 - c1__); d[3] = 1u;
        }
#endif
        tile = ntile;
        cur = nxt;
        xbuf ^= 1;
    }
#endif  // __HIP_DEVICE_COMPILE__
}

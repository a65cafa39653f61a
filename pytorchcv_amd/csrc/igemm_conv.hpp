// igemm_conv.hpp - im2col-free implicit-GEMM convolution on gfx950 MFMA with a fused BN/activation/residual
// epilogue. One kernel template serves dense 1x1 / 3x3 / kxk, strided, dilated, asymmetric-padded, stem
// (Cin <= 4, pixel-pair chunks) and block-diagonal grouped convolutions.
//
// Replaces: nn.Conv2d + nn.BatchNorm2d(eval) + activation of ConvBlock.forward
//           (reference pytorchcv/models/common/conv.py:278-286) and the unit's residual add + activation
//           (resnet.py:227-228, resnext.py:114-115, mobilenetv2.py:69-70).
//
// GEMM view (operands swapped so that channels land on accumulator rows and the epilogue writes 16-byte
// channel-contiguous NHWC pieces):
//     Y^T[ch, pixel] = sum_k Wp[ch, k] * X[pixel, k],  k = (filter row r, filter col q, input channel c)
//   * "A" operand = packed weights Wp[Cout_pad][Kpad] (K contiguous, rows in MFMA order, see pack kernel)
//   * "B" operand = NHWC activations gathered on the fly: row = output pixel m=(n,ho,wo), 16-byte chunk j of the
//     K axis = CE consecutive input channels of input pixel (ho*s-p+dy_j, wo*s-p+dx_j).
//
// Data movement: both tiles go global -> LDS with `buffer_load_dwordx4 ... lds` (LDS-DMA, no VGPR round trip).
// The per-lane SOURCE offset implements the im2col gather; a padded (out-of-image) tap is an offset beyond the
// buffer's num_records, for which the hardware writes zeros. LDS rows are 128 B (8 chunks); chunk slot s of row
// r holds K-chunk s ^ (r & 7) (swizzle applied on the source side, LDS image stays lane-linear) so that the
// `ds_read_b128` fragment reads of 16 consecutive rows are bank-conflict free.
//
// Pipeline: 2 LDS stages; per K-step one barrier: { wait DMA(t) ; barrier ; issue DMA(t+1) ; MFMA(t) }, running
// across tile boundaries (persistent blocks).
#pragma once
#include "pcv_common.hpp"

#define IGEMM_MAX_TAPS 16

struct IgemmParams {
    const void* x;          // NHWC activations
    const void* w;          // packed weights (weights region of the blob)
    const uint32_t* ktab;   // 2 dwords per K-chunk: {c0 | r<<16 | q<<20, (int16)dy | (int16)dx<<16}
    const void* res;        // residual NHWC [M, Cout_total] or null
    void* y;                // NHWC [M, Cout_total]
    const float* scale;     // [Cout_total]
    const float* shift;
    uint32_t x_bytes;       // buffer num_records for x
    uint32_t w_bytes;       // buffer num_records for w
    uint32_t y_bytes;       // buffer num_records for y (16-bit non-ragged outputs are written with range-checked stores)
    int M;                  // N*Ho*Wo
    int Cout;               // valid output channels per group-block (== Cout_total when gridDim.y == 1)
    int Cout_total;         // channel pitch of the residual (= all output channels)
    int Ypitch;             // channel pitch of y: Cout_total, or wider when y is a channel slice of a concat buffer
    int cout_blk;           // output channels per blockIdx.y step
    int cin_blk;            // input-channel offset per blockIdx.y step
    int wrows_blk;          // packed weight rows per blockIdx.y step
    FastDiv div_howo, div_wo;
    int HoWo, Wo;
    int H, W, Wpitch, Cpitch;
    int sh, sw, pt, pl;
    int nR, nQ;
    int dy[IGEMM_MAX_TAPS];
    int dx[IGEMM_MAX_TAPS];
    int nk;                 // K steps of 128 B
    int Kpad;               // packed row length in elements
    int act, post_act;
    int nPixTiles, nChTiles;
    int ngb;                // group-blocks (block-diagonal grouped convolution), 1 for dense
    int nTiles;             // nPixTiles * ngb * nChTiles
    int Cin;                // input channels per group-block (KHW == 1: chunks beyond it are zero-filled)
    int ksteps_per_tap;     // KHW == 9: K-steps per filter tap (= cin_blk / elements per K-step)
    int korder;             // KHW == 9: 0 = K ordered (r, q, slice), 1 = (r, slice, q)
    int wstat;              // weight-stationary persistent mode (nk == 1, one channel tile, one group-block)
    const float* gate;      // [N][Cout_total] fp32 or null: per-image channel gate applied between the activation and the residual
                            // add (SE blocks whose squeeze was taken upstream of this convolution, pcv_conv2d_gated_fused)
    uint32_t* ovf;          // the context's fp16 overflow counter (pcv_common.hpp, F16Guard)
};

template <int DT> struct Mma;
template <> struct Mma<PCV_BF16> {
    typedef s16x8 frag;
    static __device__ __forceinline__ f32x4 run(const frag& a, const frag& b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mma<PCV_F16> {
    typedef f16x8 frag;
    static __device__ __forceinline__ f32x4 run(const frag& a, const frag& b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mma<PCV_F32> {
    // exact-f32 MFMA (v_mfma_f32_16x16x4_f32): the 16-byte chunk held by lane group q carries k = 4q+e, and
    // MFMA e (0..3) sums element e over the four lane groups - the same k association on both operands.
    typedef f32x4 frag;
    static __device__ __forceinline__ f32x4 run(const frag& a, const frag& b, f32x4 c) {
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c, 0, 0, 0);
        return c;
    }
};

// DT: storage type of x / w / residual.  OT: storage type of y.  CB: 16-channel blocks per wave (2 or 4).
// PB: 16-pixel blocks per wave.  WC x WP: wave grid (channels x pixels).  RAGGED: Cout not a multiple of 8.
// KHW: 0 = taps from the descriptor table, 1 = 1x1 without padding (every tap valid, chunk offsets computed),
//      9 = 3x3 dilation 1 with Cin a multiple of one K-step (tap and channel offset of a K-step computed, no table).
//
// Persistent-capable: every block walks a strided list of tiles inside its XCD's contiguous tile range (one tile per
// block when the host launches as many blocks as tiles). The software pipeline runs across tile boundaries: the
// LDS-DMA loads of the next tile's first K-step are issued before this tile's last MFMAs and epilogue. The residual
// tile and scale/shift are fetched to registers BEFORE the last K-step's MFMAs so their latency hides under them.
template <int DT, int OT, int CB, int PB, int WC, int WP, bool RAGGED, int KHW>
__global__ __launch_bounds__(64 * WC * WP, 2) void igemm_conv_kernel(const IgemmParams p) {   // 2 waves per SIMD = 2 blocks per CU: caps VGPR+AGPR at 256
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass only needs the launch stub (buffer-resource types are device-only)
    constexpr int NW = WC * WP;
    constexpr int BM = 16 * CB * WC;          // channel rows per block tile
    constexpr int BP = 16 * PB * WP;          // pixel rows per block tile
    constexpr int ES = Elem<DT>::BYTES;
    constexpr int CE = 16 / ES;               // elements per 16-byte chunk
    constexpr int STAGE = (BM + BP) * 128;    // bytes per LDS stage
    constexpr int WLOADS = BM / (8 * NW);     // LDS-DMA wave-instructions per thread for the weight tile
    constexpr int XLOADS = BP / (8 * NW);
    constexpr int NPAIR = CB / 2;
    static_assert(BM % (8 * NW) == 0 && BP % (8 * NW) == 0, "tile rows must split evenly over the waves");
    typedef typename Mma<DT>::frag frag;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave / WP, wp = wave % WP;

    // ---- this block's tile list: tiles [tile, tend) of its XCD's range, stride = blocks per XCD ----------------
    const int perXcd = (p.nTiles + 7) >> 3;
    const int xcd = blockIdx.x & 7;
    const int tstride = gridDim.x >> 3;                      // host guarantees gridDim.x % 8 == 0
    int tile = xcd * perXcd + (int)(blockIdx.x >> 3);
    const int tend = min(p.nTiles, (xcd + 1) * perXcd);
    if (tile >= tend) return;

    const int lrow = lane >> 3;               // row within an 8-row DMA piece
    const int cs = (lane & 7) ^ lrow;         // K-chunk this lane fetches (source-side swizzle)
    const int nR = KHW == 9 ? 3 : (KHW == 1 ? 1 : p.nR);
    const int nQ = KHW == 9 ? 3 : (KHW == 1 ? 1 : p.nQ);
    const int fr = lane & 15, fq = lane >> 4;

    // per-tile gather state: XLOADS pixel rows + WLOADS weight rows per thread
    struct TileState {
        int rbase[XLOADS];
        uint32_t rmask[XLOADS];
        uint32_t woff[WLOADS];
        int chTile, gb, tileP0;
    };
    auto setup = [&](int t, TileState& S) {
        S.chTile = t % p.nChTiles;
        const int t2 = t / p.nChTiles;
        S.gb = t2 % p.ngb;
        S.tileP0 = (t2 / p.ngb) * BP;
#pragma unroll
        for (int i = 0; i < XLOADS; ++i) {
            const int m = S.tileP0 + 8 * (i * NW + wave) + lrow;
            uint32_t mask = 0;
            int base = 0;
            if (m < p.M) {
                const uint32_t n = fastdiv((uint32_t)m, p.div_howo);
                const uint32_t rem = (uint32_t)m - n * (uint32_t)p.HoWo;
                const uint32_t ho = fastdiv(rem, p.div_wo);
                const uint32_t wo = rem - ho * (uint32_t)p.Wo;
                const int hi0 = (int)ho * p.sh - p.pt;
                const int wi0 = (int)wo * p.sw - p.pl;
                base = (((int)n * p.H + hi0) * p.Wpitch + wi0) * p.Cpitch + S.gb * p.cin_blk;
                if constexpr (KHW == 1) {
                    mask = 0x00010001u;
                } else {
                    for (int r = 0; r < nR; ++r)
                        mask |= ((uint32_t)(hi0 + p.dy[r]) < (uint32_t)p.H ? 1u : 0u) << r;
                    for (int q = 0; q < nQ; ++q)
                        mask |= ((uint32_t)(wi0 + p.dx[q]) < (uint32_t)p.W ? 1u : 0u) << (16 + q);
                }
            }
            S.rbase[i] = base;
            S.rmask[i] = mask;
        }
#pragma unroll
        for (int i = 0; i < WLOADS; ++i) {
            const int wrow = 8 * (i * NW + wave) + lrow;
            S.woff[i] = (uint32_t)(((S.gb * p.wrows_blk + S.chTile * BM + wrow) * p.Kpad + cs * CE) * ES);
        }
    };

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);

    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);

    // 16-bit non-ragged outputs are written with range-checked buffer stores: one store instruction per (ip, j) for every
    // wave, no branches in the epilogue (tile tails / channel padding get an offset beyond num_records and are dropped).
    // (Leaving those stores in flight across the next barrier with a counted vmcnt was measured: no gain, removed.)
    constexpr bool FAST_STORE = !RAGGED && OT != PCV_F32;

    // issue the LDS-DMA loads of K-step k of tile state S into stage `buf`
    // KHW == 0: the K-chunk descriptor of a step is fetched one step ahead (`kd_next`), so that its global-load latency
    // hides under the previous step's MFMAs instead of stalling - and draining - the DMA issue.
    auto load_kdesc = [&](int k) -> u32x2 {
        return *reinterpret_cast<const u32x2*>(p.ktab + 2 * (k * 8 + cs));
    };
    u32x2 kd_next = {0u, 0u};
    // Weight-stationary mode (host sets p.wstat when a tile has ONE K-step and ONE channel tile covers all outputs,
    // i.e. every tile of this block uses the same 128-byte-per-row weight slab): the slab is loaded once into stage 0's
    // weight rows and only the activation rows are streamed afterwards - the HBM-bound 1x1 layers with Cin <= 64.
    const bool wstat = (KHW == 1) && p.wstat != 0;      // compile-time false for the 3x3 / table-driven instantiations
    bool w_loaded = false;
    auto stage = [&](const TileState& S, int k, int buf) {
        char* sbase = smem + buf * STAGE;
        if (!(wstat && w_loaded)) {
#pragma unroll
            for (int i = 0; i < WLOADS; ++i) {
                char* dst = sbase + (8 * (i * NW + wave)) * 128;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, PCV_LDS(dst), 16, S.woff[i], k * 128, 0, 0);
            }
            w_loaded = true;
        }
        int koff;
        uint32_t r, q;
        bool chunk_ok = true;
        if constexpr (KHW == 1) {
            const int c0 = (k * 8 + cs) * CE;                  // channel chunk; beyond Cin -> zero fill
            koff = c0;
            r = 0; q = 0;
            chunk_ok = c0 < p.Cin;
        } else if constexpr (KHW == 9) {
            int cslice;                                       // uniform: a K-step never straddles taps here
            if (p.korder == 0) {                              // K = (r, q, slice)
                const int tap = k / p.ksteps_per_tap;
                cslice = k - tap * p.ksteps_per_tap;
                r = (uint32_t)tap / 3u;
                q = (uint32_t)tap - 3u * r;
            } else {                                          // K = (r, slice, q): blobs packed for conv3x3_kernel
                const int grp = k / 3;
                q = (uint32_t)(k - 3 * grp);
                r = (uint32_t)(grp / p.ksteps_per_tap);
                cslice = grp - (int)r * p.ksteps_per_tap;
            }
            const int c0 = cslice * (8 * CE) + cs * CE;
            koff = ((int)r * p.Wpitch + (int)q) * p.Cpitch + c0;
        } else {
            const u32x2 kd = kd_next;
            const int c0 = (int)(kd[0] & 0xFFFFu);
            r = (kd[0] >> 16) & 15u;
            q = (kd[0] >> 20) & 15u;
            const int dy = (int)(short)(kd[1] & 0xFFFFu), dx = (int)(short)(kd[1] >> 16);
            koff = (dy * p.Wpitch + dx) * p.Cpitch + c0;
        }
#pragma unroll
        for (int i = 0; i < XLOADS; ++i) {
            char* dst = sbase + (BM + 8 * (i * NW + wave)) * 128;
            const bool ok = chunk_ok && (((S.rmask[i] >> r) & (S.rmask[i] >> (16 + q)) & 1u) != 0);
            const uint32_t voff = ok ? (uint32_t)((S.rbase[i] + koff) * ES) : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, PCV_LDS(dst), 16, voff, 0, 0, 0);
        }
    };

    // ---- fragment read addresses ------------------------------------------------------------------------
    const int swz0 = ((fq) ^ (fr & 7)) << 4;
    const int swz1 = ((fq + 4) ^ (fr & 7)) << 4;
    const int wfrag = (wc * 16 * CB + fr) * 128;
    const int xfrag = (BM + wp * 16 * PB + fr) * 128;

    f32x4 acc[CB][PB];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < CB; ++i)
#pragma unroll
            for (int j = 0; j < PB; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    };
    auto compute = [&](int buf) {
        const char* sbase = smem + buf * STAGE;
        const char* wsbase = wstat ? smem : sbase;       // weight-stationary: the slab lives in stage 0
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int swz = kk == 0 ? swz0 : swz1;
            frag a[CB], b[PB];
#pragma unroll
            for (int i = 0; i < CB; ++i) a[i] = *reinterpret_cast<const frag*>(wsbase + wfrag + i * 2048 + swz);
#pragma unroll
            for (int j = 0; j < PB; ++j) b[j] = *reinterpret_cast<const frag*>(sbase + xfrag + j * 2048 + swz);
#pragma unroll
            for (int i = 0; i < CB; ++i)
#pragma unroll
                for (int j = 0; j < PB; ++j) acc[i][j] = Mma<DT>::run(a[i], b[j], acc[i][j]);
        }
    };

    const ActClamp act = make_act(p.act), pact = make_act(p.post_act);
    // the stored value is bounded by construction (no range check needed) when the last thing applied to it is a bounded activation
    const bool bounded = act_bounded(p.post_act) || (p.post_act == PCV_ACT_NONE && p.res == nullptr && p.gate == nullptr && act_bounded(p.act));
    zero_acc();
    TileState cur, nxt;
    setup(tile, cur);
    const int nk = p.nk;
    if constexpr (KHW == 0) kd_next = load_kdesc(0);
    stage(cur, 0, 0);
    if constexpr (KHW == 0) kd_next = load_kdesc(nk > 1 ? 1 : 0);
    int buf = 0;

    while (true) {
        // ---- K-steps 0 .. nk-2: tight loop, one barrier each --------------------------------------------------
        for (int k = 0; k + 1 < nk; ++k) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // LDS-DMA completion is a vmcnt event: say so, do not leave it to the compiler
            __syncthreads();                   // DMA(k) landed for every wave; the other stage is free again
            stage(cur, k + 1, buf ^ 1);
            if constexpr (KHW == 0) kd_next = load_kdesc(k + 2 < nk ? k + 2 : 0);    // (k+2 == nk: next tile's step 0)
            compute(buf);
            buf ^= 1;
        }
        // ---- last K-step of the tile: prefetch next tile's first K-step + this tile's epilogue operands --------
        const int ntile = tile + tstride;
        const bool has_next = ntile < tend;
        if (has_next) setup(ntile, nxt);       // address math overlaps the DMA wait
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        // Packed weight row (16*i + rho) of a 64-row group holds channel 32*(i>>1) + 8*(rho>>2) + 4*(i&1) + (rho&3), so
        // lane group fq owns the 8 consecutive channels 32*ip + 8*fq .. +7 (accumulators 2ip and 2ip+1).
        const int chBlk = cur.chTile * BM + wc * 16 * CB;     // first channel (within the group-block) of this wave
        const int chGlob0 = cur.gb * p.cout_blk;
        const int mBase = cur.tileP0 + wp * 16 * PB + fr;
        float sc[NPAIR][8], sf[NPAIR][8];
        u32x4 rres[NPAIR][PB];                                  // 16-bit residual: 8 channels per (ip, j)
        f32x4 rres32[DT == PCV_F32 ? NPAIR : 1][DT == PCV_F32 ? PB : 1][2];
        if constexpr (!RAGGED) {
#pragma unroll
            for (int ip = 0; ip < NPAIR; ++ip) {
                const int ch0 = chBlk + 32 * ip + 8 * fq;
                const int chg = chGlob0 + ch0;
                f32x4 s0 = {1.f, 1.f, 1.f, 1.f}, s1 = s0, h0 = {0.f, 0.f, 0.f, 0.f}, h1 = h0;
                if (ch0 < p.Cout) {
                    if (p.scale != nullptr) {
                        s0 = *reinterpret_cast<const f32x4*>(p.scale + chg);
                        s1 = *reinterpret_cast<const f32x4*>(p.scale + chg + 4);
                    }
                    if (p.shift != nullptr) {
                        h0 = *reinterpret_cast<const f32x4*>(p.shift + chg);
                        h1 = *reinterpret_cast<const f32x4*>(p.shift + chg + 4);
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) { sc[ip][e] = s0[e]; sc[ip][4 + e] = s1[e]; sf[ip][e] = h0[e]; sf[ip][4 + e] = h1[e]; }
#pragma unroll
                for (int j = 0; j < PB; ++j) {
                    const int m = mBase + 16 * j;
                    const bool ok = p.res != nullptr && ch0 < p.Cout && m < p.M;
                    const size_t eoff = (size_t)m * p.Cout_total + chg;
                    if constexpr (DT == PCV_F32) {
                        rres32[ip][j][0] = ok ? *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.res) + eoff)
                                              : (f32x4){0.f, 0.f, 0.f, 0.f};
                        rres32[ip][j][1] = ok ? *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.res) + eoff + 4)
                                              : (f32x4){0.f, 0.f, 0.f, 0.f};
                    } else {
                        rres[ip][j] = ok ? *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint16_t*>(p.res) + eoff)
                                         : (u32x4){0u, 0u, 0u, 0u};
                    }
                }
            }
        }

        // the next tile's first K-step is requested AFTER the epilogue operands, so that waiting for those (the compiler
        // counts VMEM ops in issue order) does not also wait for this DMA
        if (has_next) stage(nxt, 0, buf ^ 1);
        if constexpr (KHW == 0) kd_next = load_kdesc(nk > 1 ? 1 : 0);

        compute(buf);

        // ---- epilogue: scale/shift -> act -> (+residual) -> post_act -> NHWC store ----------------------------
        F16Guard<OT> guard;                                     // (its scalar state lives in the epilogue only, not across the K loop)
#pragma unroll
        for (int ip = 0; ip < NPAIR; ++ip) {
            const int ch0 = chBlk + 32 * ip + 8 * fq;           // within the group-block
            if constexpr (!FAST_STORE) {
                if (ch0 >= p.Cout) continue;
            }
            const int chg = chGlob0 + ch0;                      // global channel
            if constexpr (RAGGED) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const bool ok = ch0 + e < p.Cout;
                    sc[ip][e] = ok ? (p.scale ? p.scale[chg + e] : 1.f) : 0.f;
                    sf[ip][e] = ok ? (p.shift ? p.shift[chg + e] : 0.f) : 0.f;
                }
            }
#pragma unroll
            for (int j = 0; j < PB; ++j) {
                const int m = mBase + 16 * j;
                if constexpr (!FAST_STORE) {
                    if (m >= p.M) continue;
                }
                const size_t eoff = (size_t)m * p.Cout_total + chg;          // residual element
                const size_t yoff = (size_t)m * p.Ypitch + chg;              // output element
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[2 * ip][j][e] * sc[ip][e] + sf[ip][e];
                    v[4 + e] = acc[2 * ip + 1][j][e] * sc[ip][4 + e] + sf[ip][4 + e];
                }
                apply_act8(v, act);
                if (p.gate != nullptr) {
#pragma clang fp contract(off)      // the product is rounded before the skip add here and in wpair1x1.hpp alike (bit-identical paths)
                    const uint32_t n = fastdiv((uint32_t)(m < p.M ? m : 0), p.div_howo);
                    const float* gp = p.gate + (size_t)n * p.Cout_total + chg;
                    if constexpr (RAGGED) {
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (ch0 + e < p.Cout) v[e] *= gp[e];
                    } else if (ch0 < p.Cout) {
                        const f32x4 g0 = *reinterpret_cast<const f32x4*>(gp), g1 = *reinterpret_cast<const f32x4*>(gp + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v[e] *= g0[e]; v[4 + e] *= g1[e]; }
                    }
                }
                if (p.res != nullptr) {
                    if constexpr (RAGGED) {
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (ch0 + e < p.Cout) v[e] += load_elem<DT>(p.res, eoff + e);
                    } else if constexpr (DT == PCV_F32) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v[e] += rres32[ip][j][0][e]; v[4 + e] += rres32[ip][j][1][e]; }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float lo, hi;
                            unpack2<DT>(rres[ip][j][e], lo, hi);
                            v[2 * e] += lo;
                            v[2 * e + 1] += hi;
                        }
                    }
                }
                apply_act8(v, pact);
                if (!bounded) guard.see(v);
                if constexpr (RAGGED) {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (ch0 + e < p.Cout) store_elem<OT>(p.y, yoff + e, v[e]);
                } else if constexpr (OT == PCV_F32) {
                    float* yp = reinterpret_cast<float*>(p.y) + yoff;
                    *reinterpret_cast<f32x4*>(yp) = (f32x4){v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<f32x4*>(yp + 4) = (f32x4){v[4], v[5], v[6], v[7]};
                } else {
                    u32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = pack2<OT>(v[2 * e], v[2 * e + 1]);
                    // exactly one store instruction per (ip, j) for every wave: out-of-range pieces (tile tail, channel
                    // padding) get an offset beyond num_records and are dropped by the range check
                    const bool ok = ch0 < p.Cout && m < p.M;
                    const uint32_t boff = ok ? (uint32_t)(yoff * 2) : 0x80000000u;
                    __builtin_amdgcn_raw_buffer_store_b128(o, yrsrc, boff, 0, 0);
                }
            }
        }
        guard.commit(p.ovf);
        if (!has_next) break;
        zero_acc();
        cur = nxt;
        tile = ntile;
        buf ^= 1;
    }
#endif  // __HIP_DEVICE_COMPILE__
}

// conv3x3.hpp - dense 3x3 / stride 1 / pad 1 convolution on gfx950 MFMA: the MFMA-bound layers of ResNet
// (reference conv3x3_block, pytorchcv/models/common/conv.py:340-386, used at resnet.py:49,56,120) with the same fused
// scale/shift/activation/residual epilogue as igemm_conv.hpp.
//
// What it adds over the generic implicit GEMM (whose main loop is bound by LDS-DMA issue, not by MFMA):
//   * Filter-row reuse of the activation tile. K is ordered (filter row r, 64-channel slice cs, filter column q).
//     For a fixed (r, cs) the three q taps read the SAME input pixels shifted by one: the block stages ONE activation
//     tile of BP+2 pixel rows (flat pixel range [p0-1, p0+BP+1) moved by (r-1) image rows) and the q-th tap reads it
//     at row offset q. Activation loads drop 3x; only the 2 edge columns need a per-lane mask at fragment level
//     (image-row padding is resolved at load time by the buffer range check, as in the generic kernel).
//   * 8 waves / 1 block per CU with tiles 128ch x 256px or 64ch x 512px (wave tile 64x64): each weight tile is shared
//     by 4-8 pixel waves, each activation tile by 1-2 channel waves.
//   * 3-slot weight ring + double-buffered activation tile, prefetch distance 2 K-steps, counted `s_waitcnt vmcnt(N)`
//     and raw `s_barrier` (never a drain to 0 in the steady state), running across tile boundaries (persistent).
//
// Operand layouts (LDS rows of 128 B, XOR-swizzled 16-byte chunks, conflict-free ds_read_b128 for any row shift) and
// the packed-weight row order are those of igemm_conv.hpp.
#pragma once
#include <type_traits>
#include "pcv_common.hpp"
#include "igemm_conv.hpp"     // Mma<DT>

struct Conv3Params {
    const void* x;
    const void* w;          // packed weights, K order (r, cs, q, c)
    const void* res;
    void* y;
    const float* scale;
    const float* shift;
    uint32_t x_bytes, w_bytes;
    int M;                  // N*H*W output (= input) pixels
    int H, W, C;            // input height, width, channels (= channel pitch)
    int Cout;
    FastDiv div_hw, div_w;
    int HW;
    int CS;                 // K-steps (128-byte channel slices) per tap
    int Kpad;               // packed row length in elements
    int act, post_act;
    int nChTiles, nTiles;
    int flags;              // tuning switches (bit 0: lgkmcnt(0) before every barrier, bit 1: s_setprio around MFMA clusters)
};

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// DT storage type; WC x WP wave grid (channels x pixels), wave tile 64ch x 64px. Weight ring of 3 slots (slot = q),
// activation tile double-buffered per (r, cs) group, prefetch distance 2 K-steps.
//
// Schedule of one group g = (r, cs) - straight-line, all wait counts are compile-time constants:
//   q=0:  wait vmcnt(WL)     ; barrier ; issue W(g, q=2) then X(g+1)        ; MFMA(g, 0)
//   q=1:  wait vmcnt(WL+XL)  ; barrier ; issue W(g+1, q=0)                  ; MFMA(g, 1)     [X(g+1) stays in flight]
//   q=2:  wait vmcnt(WL+XL)  ; barrier ; issue W(g+1, q=1)                  ; MFMA(g, 2)     [X(g+1) stays in flight]
// (vmcnt(N): everything older than the N youngest VMEM ops of this wave has landed; ops issued later - epilogue
// stores - only make the wait stricter.) "g+1" of a tile's last group is group 0 of the block's next tile.
template <int DT, int WC, int WP>
__global__ __launch_bounds__(64 * WC * WP, 2) void conv3x3_kernel(const Conv3Params p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int CB = 4, PB = 4, NPAIR = 2;
    constexpr int NW = WC * WP;
    constexpr int BM = 64 * WC;
    constexpr int BP = 64 * WP;
    constexpr int ES = Elem<DT>::BYTES;
    constexpr int CE = 16 / ES;
    constexpr int BKE = 8 * CE;                    // elements per K-step
    constexpr int XR = BP + 8;                     // activation tile rows (BP + 2 used)
    constexpr int XPIECES = XR / 8;
    constexpr int XL = (XPIECES + NW - 1) / NW;    // activation DMA instructions per thread per group
    constexpr int WL = BM / (8 * NW);              // weight DMA instructions per thread per step
    constexpr int WRING = 3 * BM * 128;
    constexpr int ZROW = WRING + 2 * XR * 128;     // one 128-byte row of zeros: what a padded tap reads
    static_assert(BM % (8 * NW) == 0, "weight tile must split evenly over the waves");
    typedef typename Mma<DT>::frag frag;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave / WP, wp = wave % WP;
    const int lrow = lane >> 3;
    const int cs_lane = (lane & 7) ^ lrow;         // source-side swizzle
    const int fr = lane & 15, fq = lane >> 4;

    const int perXcd = (p.nTiles + 7) >> 3;
    const int xcd = blockIdx.x & 7;
    const int tstride = gridDim.x >> 3;
    int tile = xcd * perXcd + (int)(blockIdx.x >> 3);
    const int tend = min(p.nTiles, (xcd + 1) * perXcd);
    if (tile >= tend) return;

    if (tid < 8) *reinterpret_cast<u32x4*>(smem + ZROW + 16 * tid) = (u32x4){0u, 0u, 0u, 0u};   // visible after the first barrier

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);
    const int rowBytes = p.W * p.C * ES;           // bytes per image row
    const int G = 3 * p.CS;                        // groups per tile

    struct TileState {
        uint32_t xoff[XL];       // byte offset of this thread's chunk of the CENTER pixel of activation-tile row i
        uint32_t xmask[XL];      // bit r: filter row r is inside the image for that pixel
        uint32_t woff[WL];
        int chTile, p0;
    };
    auto setup = [&](int t, TileState& S) {
        S.chTile = t % p.nChTiles;
        S.p0 = (t / p.nChTiles) * BP;
#pragma unroll
        for (int j = 0; j < XL; ++j) {
            int piece = j * NW + wave;
            piece = piece < XPIECES ? piece : XPIECES - 1;          // surplus waves repeat the last piece (same bytes)
            const int c = S.p0 - 1 + 8 * piece + lrow;               // center pixel of tile row i = 8*piece + lrow
            uint32_t mask = 0, off = 0;
            if (c >= 0 && c < p.M) {
                const uint32_t n = fastdiv((uint32_t)c, p.div_hw);
                const uint32_t rem = (uint32_t)c - n * (uint32_t)p.HW;
                const uint32_t h = fastdiv(rem, p.div_w);
                mask = (h >= 1 ? 1u : 0u) | 2u | ((int)h <= p.H - 2 ? 4u : 0u);
                off = (uint32_t)((c * p.C + cs_lane * CE) * ES);
            }
            S.xoff[j] = off;
            S.xmask[j] = mask;
        }
#pragma unroll
        for (int i = 0; i < WL; ++i) {
            const int wrow = 8 * (i * NW + wave) + lrow;
            S.woff[i] = (uint32_t)(((S.chTile * BM + wrow) * p.Kpad + cs_lane * CE) * ES);
        }
    };

    // ---- DMA issue: weights of K-step (g, q) into ring slot q; activation tile of group g into buffer xb ---------------
    auto issue_w = [&](const TileState& S, int g, int q) {
        char* wdst = smem + q * (BM * 128);
        const int soff = (g * 3 + q) * 128;
#pragma unroll
        for (int i = 0; i < WL; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, PCV_LDS(wdst + (8 * (i * NW + wave)) * 128), 16, S.woff[i], soff, 0, 0);
    };
    auto issue_x = [&](const TileState& S, int g, int xb) {
        const int r = g / p.CS;
        const int cs = g - r * p.CS;
        char* xdst = smem + WRING + xb * (XR * 128);
        const uint32_t step_off = (uint32_t)((r - 1) * rowBytes + cs * (BKE * ES));
#pragma unroll
        for (int j = 0; j < XL; ++j) {
            int piece = j * NW + wave;
            piece = piece < XPIECES ? piece : XPIECES - 1;
            const bool ok = ((S.xmask[j] >> r) & 1u) != 0;
            const uint32_t voff = ok ? S.xoff[j] + step_off : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, PCV_LDS(xdst + (8 * piece) * 128), 16, voff, 0, 0, 0);
        }
    };

    // ---- compute side ------------------------------------------------------------------------------------------------
    f32x4 acc[CB][PB];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < CB; ++i)
#pragma unroll
            for (int j = 0; j < PB; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    };
    const int wfrag = (wc * 64 + fr) * 128;
    const int xrow0 = wp * 64 + fr;                    // activation-tile row of this lane's pixel for q = 0
    uint32_t colmask_lo = 0, colmask_hi = 0;           // bit jb: pixel of block jb sits in image column 0 / W-1
    auto col_masks = [&](int p0) {
        colmask_lo = colmask_hi = 0;
#pragma unroll
        for (int jb = 0; jb < PB; ++jb) {
            const int m = p0 + wp * 64 + jb * 16 + fr;
            const uint32_t mm = (uint32_t)(m < p.M ? m : 0);
            const uint32_t n = fastdiv(mm, p.div_hw);
            const uint32_t rem = mm - n * (uint32_t)p.HW;
            const uint32_t h = fastdiv(rem, p.div_w);
            const uint32_t w = rem - h * (uint32_t)p.W;
            colmask_lo |= (w == 0u ? 1u : 0u) << jb;
            colmask_hi |= ((int)w == p.W - 1 ? 1u : 0u) << jb;
        }
    };
    auto compute = [&](auto QC, int xb) {
        constexpr int q = decltype(QC)::value;
        const char* wbase = smem + q * (BM * 128) + wfrag;
        const char* xbase = smem + WRING + xb * (XR * 128) + (xrow0 + q) * 128;
        const int rsw = (fr + q) & 7;                  // swizzle term of the shifted activation row
        if (p.flags & 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int wsw = ((fq + 4 * kk) ^ (fr & 7)) << 4;
            const int xsw = ((fq + 4 * kk) ^ rsw) << 4;
            frag a[CB], b[PB];
#pragma unroll
            for (int i = 0; i < CB; ++i) a[i] = *reinterpret_cast<const frag*>(wbase + i * 2048 + wsw);
#pragma unroll
            for (int j = 0; j < PB; ++j) {
                b[j] = *reinterpret_cast<const frag*>(xbase + j * 2048 + xsw);
                if constexpr (q != 1) {                // left/right image border: this tap reads padding
                    const uint32_t kill = q == 0 ? colmask_lo : colmask_hi;
                    if ((kill >> j) & 1u) b[j] = (frag){};
                }
            }
#pragma unroll
            for (int i = 0; i < CB; ++i)
#pragma unroll
                for (int j = 0; j < PB; ++j) acc[i][j] = Mma<DT>::run(a[i], b[j], acc[i][j]);
        }
        if (p.flags & 2) __builtin_amdgcn_s_setprio(0);
    };
    auto sync = [&](auto NC) {
        // this wave's LDS reads of the previous step are complete (the slot they read may be refilled after the barrier),
        // its DMA of the coming step has landed; then everybody's
        if (p.flags & 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wait_vmcnt<decltype(NC)::value>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;

    const ActClamp act = make_act(p.act), pact = make_act(p.post_act);
    const bool late = (p.flags & 4) != 0 && (wave & 4) != 0;
    TileState cur, nxt;
    setup(tile, cur);
    col_masks(cur.p0);
    zero_acc();
    int xb = 0;                                        // activation buffer of the group being computed
    issue_w(cur, 0, 0);
    issue_x(cur, 0, 0);
    issue_w(cur, 0, 1);

    while (true) {
        // ---- groups 0 .. G-2: the following group belongs to the same tile -----------------------------------------
        for (int g = 0; g + 1 < G; ++g) {
            // Waves w and w + 4 share a SIMD. With `late` the upper half of the block issues its DMA AFTER the step's MFMAs,
            // the lower half before: on every SIMD one wave is in its (issue-bound) DMA phase while the other feeds the MFMA
            // pipe - what two independent 4-wave blocks do by drifting apart, and one 8-wave block in lockstep cannot.
            sync(std::integral_constant<int, WL>{});
            if (!late) { issue_w(cur, g, 2); issue_x(cur, g + 1, xb ^ 1); }
            compute(I0{}, xb);
            if (late) { issue_w(cur, g, 2); issue_x(cur, g + 1, xb ^ 1); }
            sync(std::integral_constant<int, WL + XL>{});
            if (!late) issue_w(cur, g + 1, 0);
            compute(I1{}, xb);
            if (late) issue_w(cur, g + 1, 0);
            sync(std::integral_constant<int, WL + XL>{});
            if (!late) issue_w(cur, g + 1, 1);
            compute(I2{}, xb);
            if (late) issue_w(cur, g + 1, 1);
            xb ^= 1;
        }
        // ---- last group: prefetch the next tile's first group, fetch the epilogue operands, finish the tile ----------
        const int ntile = tile + tstride;
        const bool has_next = ntile < tend;
        if (has_next) setup(ntile, nxt);
        sync(std::integral_constant<int, WL>{});
        issue_w(cur, G - 1, 2);
        if (has_next) issue_x(nxt, 0, xb ^ 1);
        compute(I0{}, xb);
        if (has_next) {
            sync(std::integral_constant<int, WL + XL>{});
            issue_w(nxt, 0, 0);
        } else {
            sync(std::integral_constant<int, WL>{});
        }
        compute(I1{}, xb);
        if (has_next) {
            sync(std::integral_constant<int, WL + XL>{});
            issue_w(nxt, 0, 1);
        } else {
            sync(I0{});
        }

        const int chBlk = cur.chTile * BM + wc * 64;
        const int mBase = cur.p0 + wp * 64 + fr;
        float sc[NPAIR][8], sf[NPAIR][8];
        u32x4 rres[NPAIR][PB];
        f32x4 rres32[DT == PCV_F32 ? NPAIR : 1][DT == PCV_F32 ? PB : 1][2];
#pragma unroll
        for (int ip = 0; ip < NPAIR; ++ip) {
            const int ch0 = chBlk + 32 * ip + 8 * fq;
            f32x4 s0 = {1.f, 1.f, 1.f, 1.f}, s1 = s0, h0 = {0.f, 0.f, 0.f, 0.f}, h1 = h0;
            if (ch0 < p.Cout) {
                if (p.scale != nullptr) {
                    s0 = *reinterpret_cast<const f32x4*>(p.scale + ch0);
                    s1 = *reinterpret_cast<const f32x4*>(p.scale + ch0 + 4);
                }
                if (p.shift != nullptr) {
                    h0 = *reinterpret_cast<const f32x4*>(p.shift + ch0);
                    h1 = *reinterpret_cast<const f32x4*>(p.shift + ch0 + 4);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) { sc[ip][e] = s0[e]; sc[ip][4 + e] = s1[e]; sf[ip][e] = h0[e]; sf[ip][4 + e] = h1[e]; }
#pragma unroll
            for (int j = 0; j < PB; ++j) {
                const int m = mBase + 16 * j;
                const bool ok = p.res != nullptr && ch0 < p.Cout && m < p.M;
                const size_t eoff = (size_t)m * p.Cout + ch0;
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

        compute(I2{}, xb);
        xb ^= 1;

#pragma unroll
        for (int ip = 0; ip < NPAIR; ++ip) {
            const int ch0 = chBlk + 32 * ip + 8 * fq;
            if (ch0 >= p.Cout) continue;
#pragma unroll
            for (int j = 0; j < PB; ++j) {
                const int m = mBase + 16 * j;
                if (m >= p.M) continue;
                const size_t eoff = (size_t)m * p.Cout + ch0;
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[2 * ip][j][e] * sc[ip][e] + sf[ip][e];
                    v[4 + e] = acc[2 * ip + 1][j][e] * sc[ip][4 + e] + sf[ip][4 + e];
                }
                apply_act8(v, act);
                if (p.res != nullptr) {
                    if constexpr (DT == PCV_F32) {
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
                if constexpr (DT == PCV_F32) {
                    float* yp = reinterpret_cast<float*>(p.y) + eoff;
                    *reinterpret_cast<f32x4*>(yp) = (f32x4){v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<f32x4*>(yp + 4) = (f32x4){v[4], v[5], v[6], v[7]};
                } else {
                    u32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = pack2<DT>(v[2 * e], v[2 * e + 1]);
                    *reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(p.y) + eoff) = o;
                }
            }
        }
        if (!has_next) break;
        tile = ntile;
        cur = nxt;
        col_masks(cur.p0);
        zero_acc();
    }
#endif  // __HIP_DEVICE_COMPILE__
}

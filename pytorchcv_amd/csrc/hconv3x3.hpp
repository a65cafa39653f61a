// hconv3x3.hpp - dense 3x3 / stride 1 / pad 1 convolution with the activation tile staged ONCE for all nine taps
// (reference conv3x3_block, pytorchcv/models/common/conv.py:340-386; the MFMA-bound layers of ResNet, resnet.py:49,56,120).
//
// tests/tools/micro/mfma_lds_ceiling.cpp shows what caps the generic implicit GEMM: not MFMA, LDS bandwidth or the barrier,
// but the LDS-DMA traffic per MFMA - 8 one-KB pieces per wave per 32 MFMAs (the 128x128 tile) cap the loop at ~970 TFLOP/s,
// 4 pieces at ~1560. A 3x3 convolution re-fetches the same pixels for each of its 9 taps; here they are fetched once:
//
//   * per 64-channel slice the block stages the flat pixel range [p0 - 64, p0 + BP + 64) (BP + 128 rows of 128 B); with
//     W + 1 <= 64 every tap (r, q) of every pixel of the tile is the SAME LDS tile read at row offset (r-1)*W + (q-1).
//     Borders (image top / bottom / left / right, which in the flat order are other rows, other images or nothing) are
//     resolved at fragment level: a lane whose pixel sits on a border zeroes its B fragment for the taps that leave the image.
//   * only the weights stream per tap: BM rows x 128 B into a 3-slot ring, two steps ahead.
//   LDS-DMA per tap and block: BM/8 + (BP + 128)/72 pieces (21 for 128 x 256) instead of (BM + BP)/8 (48).
//
// Schedule of one slice (9 steps s = 3r + q, straight-line, wait counts are compile-time):
//   step s: wait vmcnt(N_s) ; barrier ; issue W(s + 2) [s == 0: then X(next slice)] ; 32 MFMA of W(s) x X shifted by tap s
//   N_s = WL, + XL for s = 1, 2 (the X tile of the next slice stays in flight). "s + 2" and "next slice" run across slice
//   and tile boundaries (persistent blocks).
// Operand layouts (128-byte rows, 16-byte chunks XOR-swizzled with row & 7, conflict-free for any row shift), the packed
// weight row order and the fused epilogue are those of igemm_conv.hpp; the K order of the blob is (r, slice, q).
#pragma once
#include <type_traits>
#include "pcv_common.hpp"
#include "igemm_conv.hpp"     // Mma<DT>

struct HConvParams {
    const void* x;
    const void* w;          // packed weights, rows of Kpad elements, K order (r, slice, q, c)
    const void* res;
    void* y;
    const float* scale;
    const float* shift;
    uint32_t x_bytes, w_bytes, y_bytes;
    int M;                  // N*H*W output (= input) pixels
    int H, W, C;            // input height, width (<= 63), channels (= channel pitch, multiple of 64 / 32 for fp32)
    int Cout;
    FastDiv div_hw, div_w;
    int HW;
    int CS;                 // 128-byte channel slices
    int Kpad;
    int act, post_act;
    int nChTiles, nTiles;
};

template <int N> __device__ __forceinline__ void hconv_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int DT, int WC, int WP>
__global__ __launch_bounds__(64 * WC * WP, 1) void hconv3x3_kernel(const HConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int CB = 4, PB = 4, NPAIR = 2;
    constexpr int NW = WC * WP;
    constexpr int BM = 64 * WC;
    constexpr int BP = 64 * WP;
    constexpr int ES = Elem<DT>::BYTES;
    constexpr int CE = 16 / ES;
    constexpr int BKE = 8 * CE;                    // elements per 128-byte slice
    constexpr int WPAD = 64;                       // pixels staged before / after the tile (needs W + 1 <= 64)
    constexpr int XR = BP + 2 * WPAD;              // activation tile rows
    constexpr int XL = XR / (8 * NW);              // activation DMA instructions per thread per slice
    constexpr int WL = BM / (8 * NW);              // weight DMA instructions per thread per step
    constexpr int WRING = 3 * BM * 128;
    static_assert(BM % (8 * NW) == 0 && XR % (8 * NW) == 0, "tiles must split evenly over the waves");
    typedef typename Mma<DT>::frag frag;

    extern __shared__ __attribute__((aligned(16))) char smem[];   // [W ring: 3 x BM rows][X: 2 x XR rows]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave / WP, wp = wave % WP;
    const int lrow = lane >> 3;
    const int cs_lane = (lane & 7) ^ lrow;         // source-side swizzle: LDS slot (lane & 7) of row lrow holds this chunk
    const int fr = lane & 15, fq = lane >> 4;

    const int perXcd = (p.nTiles + 7) >> 3;
    const int xcd = blockIdx.x & 7;
    const int tstride = gridDim.x >> 3;
    int tile = xcd * perXcd + (int)(blockIdx.x >> 3);
    const int tend = min(p.nTiles, (xcd + 1) * perXcd);
    if (tile >= tend) return;

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);

    struct TileState {
        uint32_t xoff[XL];       // byte offset of this thread's chunk of activation-tile row 8*(j*NW + wave) + lrow, slice 0
        uint32_t woff[WL];
        int chTile, p0;
    };
    auto setup = [&](int t, TileState& S) {
        S.chTile = t % p.nChTiles;
        S.p0 = (t / p.nChTiles) * BP;
#pragma unroll
        for (int j = 0; j < XL; ++j) {
            const int c = S.p0 - WPAD + 8 * (j * NW + wave) + lrow;           // flat pixel held by that tile row
            S.xoff[j] = (c >= 0 && c < p.M) ? (uint32_t)((c * p.C + cs_lane * CE) * ES) : 0x80000000u;
        }
#pragma unroll
        for (int i = 0; i < WL; ++i) {
            const int wrow = 8 * (i * NW + wave) + lrow;
            S.woff[i] = (uint32_t)(((S.chTile * BM + wrow) * p.Kpad + cs_lane * CE) * ES);
        }
    };
    // weights of (slice cs, step s = 3r + q) into ring slot s % 3; the blob's K order is (r, slice, q)
    auto issue_w = [&](const TileState& S, int cs, int s) {
        const int r = s / 3, q = s - 3 * r;
        char* wdst = smem + (s % 3) * (BM * 128);
        const int soff = ((r * p.CS + cs) * 3 + q) * 128;
#pragma unroll
        for (int i = 0; i < WL; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, PCV_LDS(wdst + (8 * (i * NW + wave)) * 128), 16, S.woff[i], soff, 0, 0);
    };
    // `live` false: nothing follows (last slice of the block's last tile) - the instructions are still issued, out of range,
    // so that the compile-time wait counts of the schedule hold
    auto issue_x = [&](const TileState& S, int cs, int xb, bool live) {
        char* xdst = smem + WRING + xb * (XR * 128);
        const int soff = cs * (BKE * ES);
#pragma unroll
        for (int j = 0; j < XL; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, PCV_LDS(xdst + (8 * (j * NW + wave)) * 128), 16,
                                                     live ? S.xoff[j] : 0x80000000u, soff, 0, 0);
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
    const int xrow0 = WPAD + wp * 64 + fr;             // tile row of this lane's pixel (block jb adds 16*jb)
    uint32_t m_top = 0, m_bot = 0, m_lo = 0, m_hi = 0; // bit jb: the pixel of block jb lies in image row 0 / H-1, column 0 / W-1
    auto border_masks = [&](int p0) {
        m_top = m_bot = m_lo = m_hi = 0;
#pragma unroll
        for (int jb = 0; jb < PB; ++jb) {
            const int m = p0 + wp * 64 + jb * 16 + fr;
            const uint32_t mm = (uint32_t)(m < p.M ? m : 0);
            const uint32_t n = fastdiv(mm, p.div_hw);
            const uint32_t rem = mm - n * (uint32_t)p.HW;
            const uint32_t h = fastdiv(rem, p.div_w);
            const uint32_t w = rem - h * (uint32_t)p.W;
            m_top |= (h == 0u ? 1u : 0u) << jb;
            m_bot |= ((int)h == p.H - 1 ? 1u : 0u) << jb;
            m_lo |= (w == 0u ? 1u : 0u) << jb;
            m_hi |= ((int)w == p.W - 1 ? 1u : 0u) << jb;
        }
    };
    auto compute = [&](auto SC, int xb) {
        constexpr int s = decltype(SC)::value;
        constexpr int r = s / 3, q = s % 3;
        const char* wbase = smem + (s % 3) * (BM * 128) + wfrag;
        const int row = xrow0 + (r - 1) * p.W + (q - 1);          // tile row this lane reads for block 0
        const char* xbase = smem + WRING + xb * (XR * 128) + row * 128;
        const int rsw = row & 7;                                   // swizzle term of that row (+16*jb does not change it)
        const uint32_t kill = (r == 0 ? m_top : 0u) | (r == 2 ? m_bot : 0u) | (q == 0 ? m_lo : 0u) | (q == 2 ? m_hi : 0u);
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
                if constexpr (s != 4) {                            // every tap but the centre can leave the image
                    if ((kill >> j) & 1u) b[j] = (frag){};
                }
            }
#pragma unroll
            for (int i = 0; i < CB; ++i)
#pragma unroll
                for (int j = 0; j < PB; ++j) acc[i][j] = Mma<DT>::run(a[i], b[j], acc[i][j]);
        }
    };
    auto sync = [&](auto NC) {
        hconv_wait_vmcnt<decltype(NC)::value>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    const ActClamp act = make_act(p.act), pact = make_act(p.post_act);
    TileState cur, nxt;
    setup(tile, cur);
    border_masks(cur.p0);
    zero_acc();
    int xb = 0;
    issue_x(cur, 0, 0, true);
    issue_w(cur, 0, 0);
    issue_w(cur, 0, 1);

    using IWL = std::integral_constant<int, WL>;
    using IWX = std::integral_constant<int, WL + XL>;

    while (true) {
        const int ntile = tile + tstride;
        const bool has_next = ntile < tend;
        if (has_next) setup(ntile, nxt);
        for (int cs = 0; cs < p.CS; ++cs) {
            const bool last = cs + 1 == p.CS;
            // (slice, tile state) that the look-ahead of this slice's late steps belongs to
            const bool la_ok = !last || has_next;
            const TileState& LA = last ? nxt : cur;
            const int la_cs = last ? 0 : cs + 1;
            // step 0: weights of step 2, then the next slice's activation tile
            sync(IWL{});
            issue_w(cur, cs, 2);
            issue_x(LA, la_cs, xb ^ 1, la_ok);
            compute(std::integral_constant<int, 0>{}, xb);
            sync(IWX{});   issue_w(cur, cs, 3);   compute(std::integral_constant<int, 1>{}, xb);
            sync(IWX{});   issue_w(cur, cs, 4);   compute(std::integral_constant<int, 2>{}, xb);
            sync(IWL{});   issue_w(cur, cs, 5);   compute(std::integral_constant<int, 3>{}, xb);
            sync(IWL{});   issue_w(cur, cs, 6);   compute(std::integral_constant<int, 4>{}, xb);
            sync(IWL{});   issue_w(cur, cs, 7);   compute(std::integral_constant<int, 5>{}, xb);
            sync(IWL{});   issue_w(cur, cs, 8);   compute(std::integral_constant<int, 6>{}, xb);
            sync(IWL{});
            if (la_ok) issue_w(LA, la_cs, 0);
            if (!last) {
                compute(std::integral_constant<int, 7>{}, xb);
                sync(IWL{});
                issue_w(LA, la_cs, 1);
                compute(std::integral_constant<int, 8>{}, xb);
                xb ^= 1;
            }
        }
        // ---- the tile's last two steps (7, 8 of its last slice) frame the epilogue operand fetch ---------------------
        const int chBlk = cur.chTile * BM + wc * 64;
        const int mBase = cur.p0 + wp * 64 + fr;
        compute(std::integral_constant<int, 7>{}, xb);
        if (has_next) { sync(IWL{}); issue_w(nxt, 0, 1); } else { sync(std::integral_constant<int, 0>{}); }

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

        compute(std::integral_constant<int, 8>{}, xb);
        xb ^= 1;

#pragma unroll
        for (int ip = 0; ip < NPAIR; ++ip) {
            const int ch0 = chBlk + 32 * ip + 8 * fq;
#pragma unroll
            for (int j = 0; j < PB; ++j) {
                const int m = mBase + 16 * j;
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
                const bool ok = ch0 < p.Cout && m < p.M;
                if constexpr (DT == PCV_F32) {
                    if (ok) {
                        float* yp = reinterpret_cast<float*>(p.y) + eoff;
                        *reinterpret_cast<f32x4*>(yp) = (f32x4){v[0], v[1], v[2], v[3]};
                        *reinterpret_cast<f32x4*>(yp + 4) = (f32x4){v[4], v[5], v[6], v[7]};
                    }
                } else {
                    u32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = pack2<DT>(v[2 * e], v[2 * e + 1]);
                    __builtin_amdgcn_raw_buffer_store_b128(o, yrsrc, ok ? (uint32_t)(eoff * ES) : 0x80000000u, 0, 0);
                }
            }
        }
        if (!has_next) break;
        tile = ntile;
        cur = nxt;
        border_masks(cur.p0);
        zero_acc();
    }
#endif  // __HIP_DEVICE_COMPILE__
}

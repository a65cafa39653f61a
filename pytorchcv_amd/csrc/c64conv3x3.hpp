// c64conv3x3.hpp - dense 3x3 / stride 1 / pad 1 convolution with 64 input and 64 output channels (ResNet-18/34 stage 1,
// the `conv2` of ResNet-50/101/152 stage 1; reference resnet.py:49,56,120 via conv3x3_block, common/conv.py:340-386), 16 bit.
//
// On the generic implicit GEMM this shape runs on the 64 x 256 tile: 40 KB of operands per K-step through the 64 B/clk
// L1 -> LDS path for 128 MFMAs (640 against 512 clocks), every pixel fetched nine times: ~620 TFLOP/s. Here nothing
// streams inside the K loop:
//   * ALL weights (64 x 576, 72 KB) live in REGISTERS for the whole persistent block: wave = 64 channels x 64 pixels, its A
//     fragments a[tap][kk][i] are 288 VGPRs of the 512-register budget of a one-wave-per-SIMD block;
//   * the activation tile is staged ONCE for all nine taps (hconv3x3.hpp's flat halo scheme: pixel range
//     [p0 - 64, p0 + 256 + 64) of the NHWC map, taps = row shifts (r-1) W + (q-1), W + 1 <= 64, borders resolved by zeroing
//     the B fragment of a lane whose pixel leaves the image), double buffered: the next tile's 48 KB arrive by LDS-DMA
//     while the 288 MFMAs of this tile run;
//   * LDS is only read for B fragments (0.25 KB per MFMA, half of what the 64x64 wave tile of the generic kernel needs).
// The residual tile (ResNet-18/34: the block's second convolution) has no registers left to wait in: it is staged by LDS-DMA
// too (32 KB, each wave its own 64 rows, issued at the tile start, read in the epilogue).
// One barrier per tile. Epilogue (BN, activation, residual, activation, 16-byte stores) as in igemm_conv.hpp, so results are
// bit-identical to the generic kernel (same K order per output element: tap-major, channel-minor, fp32 MFMA accumulation).
//
// MEASURED (batch 256, 56x56, bf16): 79-86 us = 690-750 TFLOP/s against 81-85 us for the generic 64 x 256 tile; with a residual
// 97-102 us against 108-111 us. Operand delivery is no longer the limit, instruction issue is: with ONE wave per SIMD nothing
// overlaps the wave's own non-MFMA work - 316 v_cndmask (border masks) + 226 v_accvgpr_read (A fragments parked in AGPRs) +
// 72 ds_read + 12 LDS-DMA issues + ~480 VALU of epilogue per tile serialise with the 288 MFMAs (~11 K clocks per tile against
// 4.6 K of MFMA time; with the MFMAs compiled out the kernel still takes 45 of its 90 us). The generic kernel gets that overlap
// for free from its second wave per SIMD. Kept behind `pcv_set_tuning("c64", 1)` (off by default), parity-tested.
#pragma once
#include <type_traits>
#include "pcv_common.hpp"
#include "igemm_conv.hpp"     // Mma<DT>
#include "hconv3x3.hpp"       // HConvParams

template <int N> __device__ __forceinline__ void c64_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

static constexpr int kC64Lds = 2 * (256 + 128) * 128 + 2 * 64 * 4 + 256 * 128;     // two activation tiles + scale/shift + residual tile

template <int DT, bool HASRES>
__global__ __launch_bounds__(256, 1) void c64conv3x3_kernel(const HConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int PB = 4, NW = 4, BP = 64 * NW, WPAD = 64, XR = BP + 2 * WPAD, XL = XR / (8 * NW);
    constexpr int C = 64;
    static_assert(DT != PCV_F32, "16-bit storage only");
    typedef typename Mma<DT>::frag frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [X: 2 x XR rows of 128 B][scale 64][shift 64]
    float* const tsc = reinterpret_cast<float*>(smem + 2 * XR * 128);
    float* const tsf = tsc + C;
    char* const rtile = reinterpret_cast<char*>(tsf + C);         // HASRES: [256 pixel rows][128 B], chunks XOR-swizzled with row & 7

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane >> 3;
    const int cs_lane = (lane & 7) ^ lrow;
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
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res ? p.res : p.x), 0,
                                                                           p.res ? p.y_bytes : 0u, 0x00020000);

    if (tid < C) {
        tsc[tid] = p.scale ? p.scale[tid] : 1.f;
        tsf[tid] = p.shift ? p.shift[tid] : 0.f;
    }

    // ---- all weights -> registers: packed row 16 i + fr, K = 64 s + 32 kk + 8 fq .. +8 (s = 3 r + q) ----------------------
    frag a[9][2][4];
#pragma unroll
    for (int s = 0; s < 9; ++s)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                a[s][kk][i] = __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(
                    wrsrc, (uint32_t)(((16 * i + fr) * p.Kpad + 64 * s + 32 * kk + 8 * fq) * 2), 0, 0));

    // activation tile of pixel-tile t into buffer xb: tile row 8 (j NW + wave) + lrow holds flat pixel p0 - 64 + row
    auto issue_x = [&](int t, int xb, bool live) {
        char* xdst = smem + xb * (XR * 128);
        const int p0 = t * BP;
#pragma unroll
        for (int j = 0; j < XL; ++j) {
            const int c = p0 - WPAD + 8 * (j * NW + wave) + lrow;
            const uint32_t off = (live && c >= 0 && c < p.M) ? (uint32_t)((c * C + cs_lane * 8) * 2) : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, PCV_LDS(xdst + (8 * (j * NW + wave)) * 128), 16, off, 0, 0, 0);
        }
    };
    const int xrow0 = WPAD + wave * 64 + fr;           // tile row of this lane's pixel (block jb adds 16 jb)
    const ActClamp act = make_act(p.act), pact = make_act(p.post_act);

    issue_x(tile, 0, true);
    int xb = 0;
    bool first = true;
    while (true) {
        const int ntile = tile + tstride;
        const bool has_next = ntile < tend;
        const int p0 = tile * BP;
        // bit jb: the pixel of block jb lies in image row 0 / H-1, column 0 / W-1
        uint32_t m_top = 0, m_bot = 0, m_lo = 0, m_hi = 0;
#pragma unroll
        for (int jb = 0; jb < PB; ++jb) {
            const int m = p0 + wave * 64 + jb * 16 + fr;
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
        // X(tile) landed: the only younger VMEM ops of this wave are the previous tile's stores [8]
        if (first) c64_wait_vmcnt<0>();
        else c64_wait_vmcnt<8>();
        if (first) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // scale / shift table
        first = false;
        __builtin_amdgcn_s_barrier();                  // ... for every wave; the other buffer is no longer read
        asm volatile("" ::: "memory");
        const int mBase = p0 + wave * 64 + fr;
        if constexpr (HASRES) {
            // this wave's 64 residual rows (its own pixels): 8 pieces, ahead of the next tile's activations in the queue
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int m = p0 + wave * 64 + 8 * j + lrow;
                const uint32_t off = m < p.M ? (uint32_t)((m * C + cs_lane * 8) * 2) : 0x80000000u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rrsrc, PCV_LDS(rtile + (wave * 64 + 8 * j) * 128), 16, off, 0, 0, 0);
            }
        }
        issue_x(ntile, xb ^ 1, has_next);

        f32x4 acc[4][PB];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < PB; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const char* xtile = smem + xb * (XR * 128);
        // 18 steps (tap s, K half kk) of 16 MFMAs; the B fragments of step st + 1 are read while step st multiplies (one wave
        // per SIMD: nobody else hides the LDS latency) and the scheduler may not move anything across a step boundary, which
        // keeps the live B registers at two sets (the allocator otherwise hoists reads of many taps and spills)
        frag bq[2][PB];
        auto load_b = [&](auto ST) {
            constexpr int st = decltype(ST)::value, s = st >> 1, kk = st & 1, r = s / 3, q = s % 3;
            const int row = xrow0 + (r - 1) * p.W + (q - 1);
            const char* xbase = xtile + row * 128 + ((((fq + 4 * kk) ^ (row & 7))) << 4);
#pragma unroll
            for (int j = 0; j < PB; ++j) bq[st & 1][j] = *reinterpret_cast<const frag*>(xbase + j * 2048);
        };
        auto mma_step = [&](auto ST) {
            constexpr int st = decltype(ST)::value, s = st >> 1, kk = st & 1, r = s / 3, q = s % 3;
            if constexpr (s != 4) {                    // every tap but the centre can leave the image
                const uint32_t kill = (r == 0 ? m_top : 0u) | (r == 2 ? m_bot : 0u) | (q == 0 ? m_lo : 0u) | (q == 2 ? m_hi : 0u);
#pragma unroll
                for (int j = 0; j < PB; ++j)
                    if ((kill >> j) & 1u) bq[st & 1][j] = (frag){};
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < PB; ++j) acc[i][j] = Mma<DT>::run(a[s][kk][i], bq[st & 1][j], acc[i][j]);
        };
        load_b(std::integral_constant<int, 0>{});
#define C64_STEP(ST)                                                                   \
        if constexpr ((ST) + 1 < 18) load_b(std::integral_constant<int, ((ST) + 1) % 18>{}); \
        __builtin_amdgcn_sched_barrier(0);     /* the reads go out BEFORE this step's MFMAs, not after them */ \
        mma_step(std::integral_constant<int, (ST)>{});                                 \
        __builtin_amdgcn_sched_barrier(0);
        C64_STEP(0) C64_STEP(1) C64_STEP(2) C64_STEP(3) C64_STEP(4) C64_STEP(5) C64_STEP(6) C64_STEP(7) C64_STEP(8)
        C64_STEP(9) C64_STEP(10) C64_STEP(11) C64_STEP(12) C64_STEP(13) C64_STEP(14) C64_STEP(15) C64_STEP(16) C64_STEP(17)
#undef C64_STEP

        // ---- epilogue ---------------------------------------------------------------------------------------------------
        if constexpr (HASRES) c64_wait_vmcnt<XL>();     // the residual pieces landed (the next tile's activations may still fly)
#pragma unroll
        for (int ip = 0; ip < 2; ++ip) {
            const int ch0 = 32 * ip + 8 * fq;
            const f32x4 s0 = *reinterpret_cast<const f32x4*>(tsc + ch0), s1 = *reinterpret_cast<const f32x4*>(tsc + ch0 + 4);
            const f32x4 h0 = *reinterpret_cast<const f32x4*>(tsf + ch0), h1 = *reinterpret_cast<const f32x4*>(tsf + ch0 + 4);
#pragma unroll
            for (int j = 0; j < PB; ++j) {
                const int m = mBase + 16 * j;
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[2 * ip][j][e] * s0[e] + h0[e];
                    v[4 + e] = acc[2 * ip + 1][j][e] * s1[e] + h1[e];
                }
                apply_act8(v, act);
                if constexpr (HASRES) {
                    const int row = wave * 64 + 16 * j + fr;
                    const u32x4 rr = *reinterpret_cast<const u32x4*>(rtile + row * 128 + (((4 * ip + fq) ^ (row & 7)) << 4));
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float lo, hi;
                        unpack2<DT>(rr[e], lo, hi);
                        v[2 * e] += lo;
                        v[2 * e + 1] += hi;
                    }
                }
                apply_act8(v, pact);
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = pack2<DT>(v[2 * e], v[2 * e + 1]);
                __builtin_amdgcn_raw_buffer_store_b128(o, yrsrc, m < p.M ? (uint32_t)((m * C + ch0) * 2) : 0x80000000u, 0, 0);
            }
        }
        if (!has_next) break;
        tile = ntile;
        xb ^= 1;
    }
    c64_wait_vmcnt<0>();                               // the look-ahead DMA of the last tile (issued out of range)
#endif  // __HIP_DEVICE_COMPILE__
}

// d3i_conv.hpp - dense 3x3 / stride 1 / pad 1 convolution with 256 INPUT channels on maps of up to 14 x 14 pixels (ResNet stage 3: six layers
// of ResNet-50, three to eleven of ResNet-18 / 34 / 101), gfx950 MFMA: the whole IMAGE of a tile stays in LDS, the weights never touch it.
//
// Replaces: nn.Conv2d(256 -> Cout, 3x3, stride 1, padding 1) + nn.BatchNorm2d(eval) + activation of ConvBlock.forward (reference
//           pytorchcv/models/common/conv.py:278-286) at ResBottleneck.conv2 / ResBlock.conv1, conv2 of the 14 x 14 stage (resnet.py:49,56,
//           120-127), plus the residual add + ReLU of basic-block units (resnet.py:227-228) in the epilogue. Same K order (filter row, 64-channel
//           slice, filter column), same MFMA chain per accumulator, same epilogue arithmetic as d3q / d3w / igemm: bit-identical results.
//
// Why (round 5). d3w streams these layers' weights (1.18 MB per 256 output channels) AND their activations through the LDS-DMA path, which a
// computing CU ingests at ~16 B/clk: 49.5 us of K loop for 25 us of MFMA time, on 224 of 256 CUs. The register-weight recipe (d3c, d3k) ends at
// 288 weight registers per wave. What does fit on chip is the OTHER operand: one 14 x 14 image x 256 channels is 100 KB. So here
//   * a block = one image x 256 output channels: 256 images at batch 256 = one block per CU, every CU busy;
//   * the image is staged ONCE (global -> registers -> LDS) into a 16 x 16 grid of pixel slots with a zero frame; a slot is 528 bytes
//     (256 channels + 16 bytes of padding: consecutive slots start four banks apart), so every filter tap, channel slice and K-half of a
//     pixel fragment is the SAME per-lane address plus an immediate: 13 address registers, no masks, no selects, no barrier in the K loop;
//   * wave w owns output channels 64 w .. 64 w + 63 of the tile for ALL 196 pixels: 13 pixel blocks x 4 channel blocks = 208 accumulator
//     registers (one wave per SIMD, 512-register budget). Its weights come straight from global memory (L2) into registers, as MFMA A
//     fragments in a fragment-ordered copy of the packed blob (1 KB per wave-load, fully coalesced; each weight is loaded by exactly one wave
//     of the block: 20 B/clk per CU), three K-halves ahead;
//   * per K-half a wave issues 13 `ds_read_b128` + 4 `buffer_load_dwordx4` for 52 MFMAs: one LDS read per FOUR MFMAs (d3c / d3k: one per two,
//     d3w: 0.39 per MFMA plus the DMA), and nothing of it goes through LDS-DMA.
// LDS image: slot (gy, gx) = pixel (gy - 1, gx - 1), byte 528 (16 gy + gx) + 2 c for channel c. Three more grid rows behind the image are
// allocated so that the fragment reads issued ahead of the last K-half (filter row "3") stay inside the allocation.
#pragma once
#include <type_traits>
#include <utility>
#include "pcv_common.hpp"
#include "igemm_conv.hpp"     // Mma<DT>
#include "d3q_conv.hpp"       // D3Params, d3q_sync

// CIN = 512 (ResNet stage 4: 7 x 7 maps, three layers of ResNet-50, three to five of ResNet-18 / 34): the same kernel with TWO images per block,
// stacked in one 17 x 9 grid that shares the zero row between them (153 slots of 1 040 bytes = 159 KB); 98 pixels = 7 pixel blocks per
// wave (112 accumulator registers), 144 K-halves, eight 64-channel slices.
template <int CIN_> struct D3ICfgT {
    static constexpr int THREADS = 256;
    static constexpr int CIN = CIN_, SLICES = CIN / 64;
    static constexpr int NIMG = CIN == 256 ? 1 : 2;          // images per block
    static constexpr int MAXW = CIN == 256 ? 14 : 7;         // map height and width up to MAXW
    static constexpr int GW = MAXW + 2;                      // slots per grid row
    static constexpr int IROWS = MAXW + 1;                   // grid rows from one image's first row to the next one's (a shared zero row)
    static constexpr int GROWS = NIMG * IROWS + 1;           // grid rows: 16 / 17
    static constexpr int NBLK = (NIMG * MAXW * MAXW + 15) / 16;      // pixel blocks per wave: 13 (208 >= 196) / 7 (112 >= 98)
    static constexpr int BM = 256, CW = 64;                  // output channels per block / per wave
    static constexpr int PITCH = CIN * 2 + 16;               // 528 / 1 040 B per slot
    static constexpr int ROWB = GW * PITCH;                  // 8 448 / 9 360 B per grid row
    static constexpr int NSLOT = GROWS * GW;                 // 256 / 153
    // 256: one more grid row, which the look-ahead reads of the last K-half touch (143 616 B); 512: those reads are not issued (159 120 B)
    static constexpr int LDS = NSLOT * PITCH + (CIN == 256 ? ROWB : 0);
    static constexpr int KH = 9 * SLICES * 2;                // 72 / 144 K-halves
    static constexpr int JR = KH / 3;                        // 24 / 48 per filter row
    static constexpr int PFW = 3, WRING = 4;                 // weights: K-halves of look-ahead, ring slots
    static constexpr int WBYTES = KH * 4 * 1024;             // the fragment-ordered weights of one wave (64 channels)
    static constexpr int SPT = (NSLOT * 8 + THREADS - 1) / THREADS;  // staging pieces (16 B of a 64-channel slice) per thread: 8 / 5
    // Staging schedule. Slices 0 .. LEAD - 1 are staged in the prologue; slice s >= LEAD is requested in K-half 6 (s - LEAD) of filter row 0 and
    // written to LDS SDIST K-halves later, 2 K-halves before its first fragment is requested: the HBM latency under every CU's simultaneous
    // requests (~2 us) must fit into SDIST K-halves (832 / 448 matrix-pipe cycles each), or the weight loads queued behind the staging loads
    // (vmcnt retires in order) stall the MFMAs - with SDIST = 3 that cost 1 100 cycles per slice. 256: one register buffer (the registers are
    // full), 5 K-halves; 512: two buffers, 11 K-halves.
    static constexpr int LEAD = CIN == 256 ? 2 : 3, SBUF = CIN == 256 ? 1 : 2, SDIST = CIN == 256 ? 5 : 11;
    static_assert(6 * LEAD - SDIST >= 2 && SDIST < 6 * SBUF, "a slice is in LDS before its first fragment read is issued; its buffer is free again by then");
    static_assert(JR % WRING == 0 && JR % 2 == 0, "ring slots are compile-time inside the filter-row loop");
    // byte offset of K-half j = (slice, filter column, half) of a filter row behind a window's top-left slot (j >= JR: the next row's)
    static constexpr int pimm(int j) { return (j >= JR ? ROWB : 0) + (((j % JR) % 6) >> 1) * PITCH + ((j % JR) / 6) * 128 + (j & 1) * 64; }
    static_assert(LDS <= 160 * 1024, "LDS");
    static_assert(ROWB + pimm(JR - 1) < 65536, "fragment offsets are 16-bit immediates");
};
typedef D3ICfgT<256> D3ICfg;

// Timing experiments (tests/tools/sh/kernel_variants.sh; results are WRONG with a bit set): 1 = no fragment reads in the K loop, 2 = no weight
// loads in the K loop. -DD3I_CYCLES: shader-cycle stamps per wave into p.dbg (tests/tools/d3i_cycles.py).
#ifndef D3I_DBG
#define D3I_DBG 0
#endif

#if defined(__HIP_DEVICE_COMPILE__)
// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>), in order
template <class F, int... I> __device__ __forceinline__ void d3i_unroll(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }

template <int DT, int CIN>
__device__ __forceinline__ void d3i_body(const D3Params& p, char* smem) {
    typedef D3ICfgT<CIN> G;
    typedef typename Mma<DT>::frag frag;
    typedef const __attribute__((address_space(3))) frag* lds_fptr;
    typedef __attribute__((address_space(3))) u32x4* lds_wptr;
    constexpr int NB = G::NBLK;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // output channels 64 wave .. of the tile
    const int fr = lane & 15, fq = lane >> 4;
    const uint32_t lds0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(size_t)PCV_LDS(smem));
    const int chTile = (int)blockIdx.x % p.nChTiles, n = (int)blockIdx.x / p.nChTiles * G::NIMG;      // first image of the block
    const int W = p.W, HW = p.HW;
    const int NP = (n + G::NIMG <= p.M / HW ? G::NIMG : 1) * HW;      // pixels of the block (an odd batch: the last block has one image)
#ifdef D3I_CYCLES
    uint64_t cyc__[6];
    cyc__[0] = __builtin_amdgcn_s_memtime();
    const uint64_t rt0__ = __builtin_amdgcn_s_memrealtime();        // 100 MHz
#endif

    const uint32_t imgBytes = (uint32_t)(HW * G::CIN * 2);
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.x)) + (size_t)n * imgBytes, 0,
                                                                           (uint32_t)(NP * G::CIN * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    const bool has_res = p.res != nullptr;
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res), 0, has_res ? p.res_bytes : 0u, 0x00020000);

    // ---- weights: A fragments of this wave's 64 channels, K-half kh: 4 x 1 KB at (chTile * 4 + wave) * WBYTES + kh * 4096 ----
    frag Wf[G::WRING][4];
    // (a wave whose 64 channels lie behind Cout - a ragged last channel tile - loads nothing: its per-lane offset is out of range by itself.
    // The scalar offset that selects the wave's fragments is not part of the descriptor's range check.)
    const bool wok = (chTile * 4 + wave) * G::CW < p.Cout;
    const uint32_t wlane = wok ? (uint32_t)(lane * 16) : 0x80000000u;
    const uint32_t wbase = wok ? (uint32_t)((chTile * 4 + wave) * G::WBYTES) : 0u;
    auto wload = [&](int kh, auto SLc) __attribute__((always_inline)) {
        constexpr int sl = decltype(SLc)::value;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane + (uint32_t)(nb * 1024), wbase + (uint32_t)(kh * 4096), 0);
            Wf[sl][nb] = __builtin_bit_cast(frag, v);
        }
    };
    wload(0, std::integral_constant<int, 0>{});
    wload(1, std::integral_constant<int, 1>{});
    wload(2, std::integral_constant<int, 2>{});
    static_assert(G::PFW == 3, "the three loads above");

    // ---- the image(s): NSLOT slots x 16-byte pieces, SPT per thread and 64-channel slice. Frame slots and slots outside an H x W map smaller
    // than the grid are out of range: zeros. Slice 0 is staged here; the others arrive under the MFMAs of filter row 0, whose K-halves
    // 6 g .. 6 g + 5 read slice g. ----
    u32x4 sb[G::SBUF][G::SPT];
    uint32_t soff[CIN == 256 ? 1 : G::SPT];
    uint32_t slds;
    const int sc8 = tid & 7;
    if constexpr (CIN == 256) {
        // a thread keeps its grid column gx = (tid >> 3) & 15, its grid row is 2 i + (tid >> 7). Grid rows above / below the map are out of
        // range by themselves (h = -1 wraps below zero, h >= H lies behind the image's num_records); a thread in a frame column starts at
        // 2^31 and stays out of range: one base register per thread, the same arithmetic for every piece
        const int sgx = (tid >> 3) & 15, sgy0 = tid >> 7;
        soff[0] = (unsigned)(sgx - 1) < (unsigned)W ? (uint32_t)((((sgy0 - 1) * W + sgx - 1) * G::CIN + sc8 * 8) * 2) : 0x80000000u;
        slds = lds0 + (uint32_t)((sgy0 * G::GW + sgx) * G::PITCH + sc8 * 16);
    } else {
        // piece i of a thread: slot 32 i + (tid >> 3) of the 17 x 9 grid (slots 153 .. 159 do not exist: nothing is written)
#pragma unroll
        for (int i = 0; i < G::SPT; ++i) {
            const int slot = 32 * i + (tid >> 3), gy = slot / G::GW, gx = slot - gy * G::GW;
            const int img = gy >= G::IROWS + 1 ? 1 : 0, h = gy - 1 - img * G::IROWS, w = gx - 1;
            soff[i] = ((unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)W) ? (uint32_t)(((img * HW + h * W + w) * G::CIN + sc8 * 8) * 2) : 0x80000000u;
        }
        slds = lds0 + (uint32_t)((tid >> 3) * G::PITCH + sc8 * 16);
    }
    const uint32_t srow2 = (uint32_t)(2 * W * G::CIN * 2);         // (256: two map rows)
    auto sload = [&](int g, u32x4 (&buf)[G::SPT]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < G::SPT; ++i)
            buf[i] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (CIN == 256 ? soff[0] + (uint32_t)i * srow2 : soff[CIN == 256 ? 0 : i]) + (uint32_t)(g * 128), 0, 0);
    };
    auto swrite = [&](int g, const u32x4 (&buf)[G::SPT]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < G::SPT; ++i) {
            if constexpr (CIN == 256) {
                *reinterpret_cast<lds_wptr>((size_t)(slds + (uint32_t)(2 * i * G::ROWB + g * 128))) = buf[i];
            } else {
                if (32 * i + (tid >> 3) < G::NSLOT) *reinterpret_cast<lds_wptr>((size_t)(slds + (uint32_t)(32 * i * G::PITCH + g * 128))) = buf[i];
            }
        }
    };
    {
        u32x4 pb[G::LEAD][G::SPT];                                // (the prologue's own registers: every slice in flight at once)
#pragma unroll
        for (int g = 0; g < G::LEAD; ++g) sload(g, pb[g]);
#pragma unroll
        for (int g = 0; g < G::LEAD; ++g) swrite(g, pb[g]);
    }

    // ---- fragment addresses: block mb's lane pixel p = 16 mb + fr (image p / HW, its pixel p % HW) at the window's top-left slot (filter row
    // r, column q: + (GW r + q) slots); pixels behind the block's images read slot 0 (zeros; never stored) ----
    uint32_t ab[NB];
#pragma unroll
    for (int mb = 0; mb < NB; ++mb) {
        const int pix = 16 * mb + fr;
        const int img = (G::NIMG > 1 && pix >= HW) ? 1 : 0, loc = pix - img * HW;
        const int py = (int)fastdiv((uint32_t)loc, p.div_w), px = loc - py * W;
        ab[mb] = lds0 + (uint32_t)(pix < NP ? ((img * G::IROWS + py) * G::GW + px) * G::PITCH : 0) + (uint32_t)(fq * 16);
    }

    f32x4 acc[NB][4];
#pragma unroll
    for (int mb = 0; mb < NB; ++mb)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    __builtin_amdgcn_s_waitcnt(0x0070);                           // vmcnt(0) lgkmcnt(0): the image is in LDS
    d3q_sync();
#ifdef D3I_CYCLES
    cyc__[1] = __builtin_amdgcn_s_memtime();
#endif

    // ---- K loop: filter row r outer (run-time: + one grid row), the row's 24 K-halves j = (slice, column q, half) unrolled. During K-half j
    // the wave requests the pixel fragments of K-half j + 1 (other register set) and the weights of K-half j + 3 (ring of four). ----
    frag Pf[2][NB];
    auto pread = [&](auto Jc, auto MBc) __attribute__((always_inline)) {
        constexpr int j = decltype(Jc)::value, mb = decltype(MBc)::value;
        Pf[j & 1][mb] = *reinterpret_cast<lds_fptr>((size_t)ab[mb] + (size_t)G::pimm(j));
    };
    auto preads = [&](auto Jc) __attribute__((always_inline)) {
        d3i_unroll([&](auto MBc) __attribute__((always_inline)) { pread(Jc, MBc); }, std::make_integer_sequence<int, NB>{});
    };
    preads(std::integral_constant<int, 0>{});
    // BN constants of the wave's two channel-block pairs: requested in the last three K-halves, in place of the weight look-ahead that has
    // nothing left to fetch
    f32x4 es[2][2], eh[2][2];
    const int chw = chTile * G::BM + wave * G::CW + 8 * fq;          // this lane's channels of pair g: chw + 32 g .. + 7
    auto step = [&](auto Rc, auto Jc) __attribute__((always_inline)) {
        constexpr int r = decltype(Rc)::value, j = decltype(Jc)::value, kh = r * G::JR + j;
        // filter row 0: slice sf is requested in this K-half, slice sw (requested SDIST K-halves ago) is written behind it and visible after the barrier
        constexpr int sf = (r == 0 && j % 6 == 0 && j / 6 + G::LEAD < G::SLICES) ? j / 6 + G::LEAD : 0;
        constexpr int sw = (r == 0 && j >= G::SDIST && (j - G::SDIST) % 6 == 0 && (j - G::SDIST) / 6 + G::LEAD < G::SLICES) ? (j - G::SDIST) / 6 + G::LEAD : 0;
        constexpr bool ahead = CIN == 256 || kh + 1 < G::KH;          // (512: no grid row behind the last one for the last K-half's look-ahead reads)
        // (a K-half that writes a slice: the writes come FIRST in program order. LDS writes and reads may alias as far as the compiler knows, so
        // their order is kept - with the reads first no write could stand between two of them, the issue pattern below had no solution and
        // the scheduler fell back to MFMAs | reads | writes | barrier in a row: 1 100 cycles per slice)
        if constexpr (sw > 0) swrite(sw, sb[sw % G::SBUF]);
        if constexpr ((D3I_DBG & 1) == 0 && ahead)
            preads(std::integral_constant<int, j + 1>{});
        if constexpr (kh + G::PFW < G::KH) {
            if constexpr ((D3I_DBG & 2) == 0) wload(kh + G::PFW, std::integral_constant<int, (j + G::PFW) % G::WRING>{});
        } else if constexpr (kh + G::PFW == G::KH) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int chl = chw + 32 * g < p.Cout ? chw + 32 * g : 0;                // pad channels: any valid entry (never stored)
                es[g][0] = *reinterpret_cast<const f32x4*>(p.scale + chl); es[g][1] = *reinterpret_cast<const f32x4*>(p.scale + chl + 4);
                eh[g][0] = *reinterpret_cast<const f32x4*>(p.shift + chl); eh[g][1] = *reinterpret_cast<const f32x4*>(p.shift + chl + 4);
            }
        }
        if constexpr (sf > 0) sload(sf, sb[sf % G::SBUF]);
#pragma unroll
        for (int mb = 0; mb < NB; ++mb)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
                acc[mb][nb] = Mma<DT>::run(Wf[(D3I_DBG & 2) ? 0 : j % G::WRING][nb], Pf[(D3I_DBG & 1) ? 0 : (j & 1)][mb], acc[mb][nb]);
        // issue order: four MFMAs, one fragment read; a weight load behind every third read; a staging load behind each of the first SPT.
        // A K-half that writes a slice: the SPT writes one by one, then the NB reads over the remaining gaps.
        d3i_unroll([&](auto MBc) __attribute__((always_inline)) {
            constexpr int mb = decltype(MBc)::value;
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            if constexpr (sw == 0) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            } else if constexpr (mb < G::SPT) {
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            } else {
                constexpr int RG = NB - G::SPT;
                __builtin_amdgcn_sched_group_barrier(0x100, (NB * (mb - G::SPT + 1)) / RG - (NB * (mb - G::SPT)) / RG, 0);
            }
            if constexpr ((mb + 1) * 4 / NB != mb * 4 / NB) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            if constexpr (sf > 0 && mb < G::SPT) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }, std::make_integer_sequence<int, NB>{});
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (sw > 0) {                                       // every wave's pieces of slice sw are in LDS before anybody requests its fragments (two K-halves on)
            __builtin_amdgcn_s_waitcnt(0xc07f);                       // lgkmcnt(0)
            d3q_sync();
        }
    };
    auto row = [&](auto Rc) __attribute__((always_inline)) {
        d3i_unroll([&](auto Jc) __attribute__((always_inline)) { step(Rc, Jc); }, std::make_integer_sequence<int, G::JR>{});
#pragma unroll
        for (int mb = 0; mb < NB; ++mb) ab[mb] += (uint32_t)G::ROWB;
#ifdef D3I_CYCLES
        cyc__[2 + decltype(Rc)::value] = __builtin_amdgcn_s_memtime();
#endif
    };
    row(std::integral_constant<int, 0>{});
    row(std::integral_constant<int, 1>{});
    row(std::integral_constant<int, 2>{});

    // ---- epilogue: BN, activation, skip tensor, 16-byte stores. Lane (fr, fq) of channel-block pair g holds channels c0 .. c0 + 7 of pixel
    // 16 mb + fr: blocks 2 g (c0 + 0..3) and 2 g + 1 (c0 + 4..7) in the packed row order. ----
    const float alo = (p.act == PCV_ACT_RELU || p.act == PCV_ACT_RELU6) ? 0.f : -INFINITY, ahi = p.act == PCV_ACT_RELU6 ? 6.f : INFINITY;
    const float plo = (p.post_act == PCV_ACT_RELU || p.post_act == PCV_ACT_RELU6) ? 0.f : -INFINITY, phi = p.post_act == PCV_ACT_RELU6 ? 6.f : INFINITY;
    const float clo = alo > plo ? alo : plo, chi = ahi < phi ? ahi : phi;
    const int mImg = n * HW;
    F16Guard<DT> guard;
    // the skip tensor's 26 pieces of this lane, all requested up front (the fragment and weight registers are free now): one exposed HBM
    // latency instead of two, under the first half's BN arithmetic
    u32x4 rr[2][NB];
    if (has_res) {
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int mb = 0; mb < NB; ++mb) {
                const int pix = 16 * mb + fr, ch0 = chw + 32 * g;
                const uint32_t roff = (ch0 < p.Cout && pix < NP) ? (uint32_t)(((mImg + pix) * p.Cout + ch0) * 2) : 0x80000000u;
                rr[g][mb] = __builtin_amdgcn_raw_buffer_load_b128(rrsrc, roff, 0, 0);
            }
    }
    // ROc: the ResNet forms - ReLU only (no skip tensor), or no activation + skip tensor + ReLU - as one v_maximum3 per value; the general form
    // clamps to launch-uniform bounds (two instructions per clamp, infinite bounds included). Same values either way.
    auto half = [&](auto HRc, auto ROc, auto Gc) __attribute__((always_inline)) {
        constexpr bool HR = decltype(HRc)::value, RO = decltype(ROc)::value;
        constexpr int g = decltype(Gc)::value;
        const int ch0 = chw + 32 * g;
        const bool chok = ch0 < p.Cout;
        const f32x4 es0 = es[g][0], es1 = es[g][1], eh0 = eh[g][0], eh1 = eh[g][1];
#pragma unroll
        for (int mb = 0; mb < NB; ++mb) {
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int hf = e >> 1, k0 = 2 * (e & 1);
                const f32x4& a = acc[mb][2 * g + hf];
                const f32x4& sc = hf == 0 ? es0 : es1;
                const f32x4& sh = hf == 0 ? eh0 : eh1;
                float v0 = a[k0] * sc[k0] + sh[k0], v1 = a[k0 + 1] * sc[k0 + 1] + sh[k0 + 1];
                if constexpr (RO) {
                    if constexpr (HR) {
                        float lo, hi;
                        unpack2<DT>(rr[g][mb][e], lo, hi);
                        v0 += lo;
                        v1 += hi;
                    }
                    v0 = __builtin_elementwise_maximum(v0, 0.f);
                    v1 = __builtin_elementwise_maximum(v1, 0.f);
                } else if constexpr (HR) {
                    v0 = __builtin_elementwise_minimum(__builtin_elementwise_maximum(v0, alo), ahi);
                    v1 = __builtin_elementwise_minimum(__builtin_elementwise_maximum(v1, alo), ahi);
                    float lo, hi;
                    unpack2<DT>(rr[g][mb][e], lo, hi);
                    v0 += lo;
                    v1 += hi;
                    v0 = __builtin_elementwise_minimum(__builtin_elementwise_maximum(v0, plo), phi);
                    v1 = __builtin_elementwise_minimum(__builtin_elementwise_maximum(v1, plo), phi);
                } else {
                    v0 = __builtin_elementwise_minimum(__builtin_elementwise_maximum(v0, clo), chi);
                    v1 = __builtin_elementwise_minimum(__builtin_elementwise_maximum(v1, clo), chi);
                }
                guard.see2(v0, v1);
                o[e] = pack2<DT>(v0, v1);
            }
            const int pix = 16 * mb + fr;
            const uint32_t boff = (chok && pix < NP) ? (uint32_t)(((mImg + pix) * p.Ypitch + ch0) * 2) : 0x80000000u;     // (the host keeps y below 2 GiB)
            __builtin_amdgcn_raw_buffer_store_b128(o, yrsrc, boff, 0, 0);
        }
    };
    const bool relu_only = has_res ? (p.act == PCV_ACT_NONE && p.post_act == PCV_ACT_RELU)
                                   : ((p.act == PCV_ACT_RELU && p.post_act <= PCV_ACT_RELU) || (p.act == PCV_ACT_NONE && p.post_act == PCV_ACT_RELU));
    auto both = [&](auto HRc, auto ROc) __attribute__((always_inline)) {
        half(HRc, ROc, std::integral_constant<int, 0>{});
        half(HRc, ROc, std::integral_constant<int, 1>{});
    };
    if (has_res) { if (relu_only) both(std::true_type{}, std::true_type{}); else both(std::true_type{}, std::false_type{}); }
    else { if (relu_only) both(std::false_type{}, std::true_type{}); else both(std::false_type{}, std::false_type{}); }
    guard.commit(p.ovf);
#ifdef D3I_CYCLES
    __builtin_amdgcn_s_waitcnt(0x0070);
    cyc__[5] = __builtin_amdgcn_s_memtime();
    if (p.dbg != nullptr && lane == 0) {
        uint32_t* d = p.dbg + (blockIdx.x * 4 + wave) * 8;
        for (int i = 0; i < 5; ++i) d[i] = (uint32_t)(cyc__[i + 1] - cyc__[i]);
        d[5] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - rt0__); d[6] = (uint32_t)rt0__; d[7] = 1u;
    }
#endif
}
#endif  // __HIP_DEVICE_COMPILE__

template <int DT, int CIN>
__global__ __launch_bounds__(256, 1) void d3i_kernel(const D3Params p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    d3i_body<DT, CIN>(p, smem);
#endif  // __HIP_DEVICE_COMPILE__
}

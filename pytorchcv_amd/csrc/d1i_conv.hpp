// d1i_conv.hpp - 1x1 / stride 1 convolution with MANY input channels (1024 / 2048: the K-heavy pointwise layers of the bottleneck
// networks - ResNeXt-101's 1024 -> 512 at 14 x 14, the 2048-channel layers at 7 x 7; the template also builds for 512), gfx950 MFMA: d3i_conv.hpp's loop with
// the activation operand STREAMED through LDS instead of resident in it.
//
// Replaces: nn.Conv2d(Cin -> Cout, 1x1) + nn.BatchNorm2d(eval) + activation of ConvBlock.forward (reference pytorchcv/models/common/conv.py:
//           278-286) at ResBottleneck / ResNeXtBottleneck conv1 and conv3 (resnet.py:108-127, resnext.py:56-75), plus the residual add + ReLU of
//           the unit (resnet.py:227-228) in the epilogue. Same K order (input channels), same MFMA chain per accumulator, same epilogue
//           arithmetic as d3q's 1x1 mode / p1r / igemm: bit-identical results.
//
// Why (round 5). d3q's 1x1 mode streams weights AND activations through LDS-DMA (a computing CU ingests ~16 B/clk of it): 69 us for the
// 52.6 GFLOP of 1024 -> 512 at 14 x 14, batch 256; p1r keeps the weights in registers, which ends at 512 input channels x 32 output channels
// per wave. Here, as in d3i_kernel:
//   * a block = 208 consecutive pixels (13 pixel blocks) x 256 output channels; wave w owns channels 64 w .. 64 w + 63 for all of them
//     (208 accumulator registers, one wave per SIMD). Its weights come straight from L2 into registers as MFMA A fragments, from the
//     fragment-ordered copy of the packed blob (pack_d3i_kernel), three K-halves ahead: 20 B/clk per CU, nothing of it through LDS;
//   * the pixels' input channels pass through LDS in 64-channel slices: a ring of four slices, laid out as d3i's image - one 528-byte row
//     per pixel (4 x 128 bytes + 16 bytes of padding), slice s in column s % 4 - so a fragment is a per-lane address + an immediate.
//     Slices travel global -> registers -> LDS: requested four slices ahead into one of two register sets, written two slices later, read
//     two slices after that. One barrier per slice, placed in the MIDDLE of the slice (behind its first K-half), where the writes it
//     publishes are a whole K-half old and the reads it protects are two slices away: no drain, a counted lgkmcnt;
//   * per K-half a wave issues 13 ds_read_b128 + 4 weight loads (+ 7 ds_write_b128 and 7 activation loads every other K-half) for 52 MFMAs.
#pragma once
#include <type_traits>
#include <utility>
#include "pcv_common.hpp"
#include "igemm_conv.hpp"     // Mma<DT>
#include "d3q_conv.hpp"       // D3Params, d3q_sync
#include "d3i_conv.hpp"       // d3i_unroll

template <int CIN_> struct D1ICfgT {
    static constexpr int THREADS = 256;
    static constexpr int CIN = CIN_, SLICES = CIN / 64;
    static constexpr int NBLK = 13, BP = 16 * NBLK;          // 208 pixels per block
    static constexpr int BM = 256, CW = 64;                  // output channels per block / per wave
    static constexpr int NS = 4;                             // ring: slices in LDS
    static constexpr int PITCH = NS * 128 + 16;              // 528 B per pixel
    static constexpr int SPT = (BP * 8 + THREADS - 1) / THREADS;     // staging pieces (16 B) per thread and slice: 7
    static constexpr int LDS = 32 * SPT * PITCH;             // 118 272 B: 224 rows (the pieces of rows 208 .. 223 are staged like the others and never read)
    static constexpr int KH = 2 * SLICES;                    // K-halves
    static constexpr int PFW = 3, WRING = 4;                 // weights: K-halves of look-ahead, ring slots
    static constexpr int WBYTES = KH * 4 * 1024;             // the fragment-ordered weights of one wave (64 channels)
    static_assert(SLICES % NS == 0 && SLICES >= 2 * NS, "ring columns and register sets are compile-time; the prologue fills four slices");
    static_assert(KH % WRING == 0, "weight ring slots are compile-time");
};

// Timing experiments (tests/tools/sh/kernel_variants.sh; results are WRONG with a bit set): 1 = no fragment reads in the K loop, 2 = no weight
// loads in the K loop, 4 = no activation staging in the K loop, 8 = no barriers in the K loop, 16 = every block stages the first tile's pixels
// (activations from L2). -DD1I_CYCLES: shader-cycle stamps per wave (tests/tools/d1i_cycles.py).
// Measured (round 5, batch 256, bf16): 1024 -> 512 at 14 x 14: 27.9 us per block = 9.1 K cycles of prologue + 32.2 K of K loop (26.6 K of
// MFMA) + 8.6 K of epilogue, two rounds of blocks: 64 us against 69 - 74 on d3q's 1x1 mode; with the activations from L2 (bit 16) 55.6 us -
// vmcnt retires in order, so every weight load (an L2 hit) issued behind a slice's activation loads waits for HBM. Tried against that, both
// slower (K loop 35.0 K cycles): one fragment register set refilled behind its last reader + a five-K-half weight look-ahead; a 16-entry
// fragment ring (program order MFMAs | write | read per pixel block) + the same look-ahead. Also tried and dropped: a PERSISTENT form whose slice
// stream runs across tile boundaries (the last slices of a tile request, write and publish the first slices of the next: no prologue after a
// block's first tile; bit-identical on resident and 8-block grids) - 67 against 64-68 us, 1024 -> 256 39.9 against 36.4: the two slice register
// sets that must survive the epilogue push the register allocator over 256 + 48 spare AGPRs, and what it spills are slice registers whose loads
// have just been issued (load, s_waitcnt vmcnt(0), scratch store: a wait for HBM per tile).
#ifndef D1I_DBG
#define D1I_DBG 0
#endif

#if defined(__HIP_DEVICE_COMPILE__)
template <int DT, int CIN>
__device__ __forceinline__ void d1i_body(const D3Params& p, char* smem) {
    typedef D1ICfgT<CIN> G;
    typedef typename Mma<DT>::frag frag;
    typedef const __attribute__((address_space(3))) frag* lds_fptr;
    typedef __attribute__((address_space(3))) u32x4* lds_wptr;
    constexpr int NB = G::NBLK;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // output channels 64 wave .. of the tile
    const int fr = lane & 15, fq = lane >> 4;
    const uint32_t lds0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(size_t)PCV_LDS(smem));
    const int chTile = (int)blockIdx.x % p.nChTiles, m0 = (int)blockIdx.x / p.nChTiles * G::BP;     // (channel tiles of one pixel tile run side by side: its slices come from L2)
#ifdef D1I_CYCLES
    uint64_t cyc__[4];
    cyc__[0] = __builtin_amdgcn_s_memtime();
    const uint64_t rt0__ = __builtin_amdgcn_s_memrealtime();        // 100 MHz
#endif

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
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
    d3i_unroll([&](auto KHc) __attribute__((always_inline)) { wload(decltype(KHc)::value, KHc); }, std::make_integer_sequence<int, G::PFW>{});

    // ---- activations: piece i of a thread = pixel 32 i + (tid >> 3) of the tile, 16-byte chunk tid & 7 of the slice (rows 208 .. 223 belong
    // to the next tile and are never read; pixels behind the tensor are out of range: zeros) ----
    const int spix = tid >> 3, sc8 = tid & 7;
    const uint32_t sbase = (uint32_t)(((((D1I_DBG & 16) ? 0 : m0) + spix) * G::CIN + sc8 * 8) * 2);       // (the host keeps x below 2 GiB; dbg 16: every block the first tile's pixels - L2 hits)
    const uint32_t srow = (uint32_t)(32 * G::CIN * 2);
    const uint32_t slds = lds0 + (uint32_t)(spix * G::PITCH + sc8 * 16);
    auto sload = [&](int s, u32x4 (&buf)[G::SPT]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < G::SPT; ++i) buf[i] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, sbase + (uint32_t)i * srow + (uint32_t)(s * 128), 0, 0);
    };
    auto swrite = [&](int s, const u32x4 (&buf)[G::SPT]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < G::SPT; ++i) *reinterpret_cast<lds_wptr>((size_t)(slds + (uint32_t)(32 * i * G::PITCH + (s % G::NS) * 128))) = buf[i];
    };
    // slices 0, 1 to LDS; slices 2, 3 stay in the two register sets (set s & 1 holds slice s + 2 on entry to slice s)
    u32x4 sb[2][G::SPT];
    {
        u32x4 pb[2][G::SPT];
        sload(0, pb[0]);
        sload(1, pb[1]);
        sload(2, sb[0]);
        sload(3, sb[1]);
        swrite(0, pb[0]);
        swrite(1, pb[1]);
    }

    // ---- fragment addresses: block mb's lane pixel 16 mb + fr; slice column and K-half are immediates ----
    uint32_t ab[NB];
#pragma unroll
    for (int mb = 0; mb < NB; ++mb) ab[mb] = lds0 + (uint32_t)((16 * mb + fr) * G::PITCH + fq * 16);

    f32x4 acc[NB][4];
#pragma unroll
    for (int mb = 0; mb < NB; ++mb)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // slices 0, 1 are in LDS (the loads of slices 2, 3 stay in flight)
    d3q_sync();
#ifdef D1I_CYCLES
    cyc__[1] = __builtin_amdgcn_s_memtime();
#endif

    // ---- K loop: K-half kh = 2 s + h of slice s, fully unrolled. During K-half kh the wave requests the fragments of K-half kh + 1 (other
    // register set) and the weights of K-half kh + 3; during the FIRST K-half of slice s it also writes slice s + 2 (register set s & 1)
    // into ring column (s + 2) % 4 - last read in slice s - 2, which every wave left before the barrier of slice s - 1 - requests slice
    // s + 4 into the same registers, and ends at the slice's barrier, which publishes slice s + 1 (written a slice ago). ----
    frag Pf[2][NB];
    auto pread = [&](auto KHc, auto MBc) __attribute__((always_inline)) {
        constexpr int kh = decltype(KHc)::value, mb = decltype(MBc)::value;
        Pf[kh & 1][mb] = *reinterpret_cast<lds_fptr>((size_t)ab[mb] + (size_t)(((kh >> 1) % G::NS) * 128 + (kh & 1) * 64));
    };
    auto preads = [&](auto KHc) __attribute__((always_inline)) {
        d3i_unroll([&](auto MBc) __attribute__((always_inline)) { pread(KHc, MBc); }, std::make_integer_sequence<int, NB>{});
    };
    preads(std::integral_constant<int, 0>{});
    // BN constants of the wave's two channel-block pairs: requested in place of the weight look-ahead that has nothing left to fetch
    f32x4 es[2][2], eh[2][2];
    const int chw = chTile * G::BM + wave * G::CW + 8 * fq;          // this lane's channels of pair g: chw + 32 g .. + 7
    auto step = [&](auto KHc) __attribute__((always_inline)) {
        constexpr int kh = decltype(KHc)::value, s = kh >> 1;
        constexpr bool first = (kh & 1) == 0;
        constexpr bool wr = first && s + 2 < G::SLICES && (D1I_DBG & 4) == 0;       // write slice s + 2
        constexpr bool ld = first && s + 4 < G::SLICES && (D1I_DBG & 4) == 0;       // request slice s + 4
        // (a K-half that writes a slice: the writes come first in program order, d3i_conv.hpp)
        if constexpr (wr) swrite(s + 2, sb[s & 1]);
        if constexpr ((D1I_DBG & 1) == 0 && kh + 1 < G::KH) preads(std::integral_constant<int, kh + 1>{});
        if constexpr (kh + G::PFW < G::KH) {
            if constexpr ((D1I_DBG & 2) == 0) wload(kh + G::PFW, std::integral_constant<int, (kh + G::PFW) % G::WRING>{});
        } else if constexpr (kh + G::PFW == G::KH) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int chl = chw + 32 * g < p.Cout ? chw + 32 * g : 0;                // pad channels: any valid entry (never stored)
                es[g][0] = *reinterpret_cast<const f32x4*>(p.scale + chl); es[g][1] = *reinterpret_cast<const f32x4*>(p.scale + chl + 4);
                eh[g][0] = *reinterpret_cast<const f32x4*>(p.shift + chl); eh[g][1] = *reinterpret_cast<const f32x4*>(p.shift + chl + 4);
            }
        }
        if constexpr (ld) sload(s + 4, sb[s & 1]);
#pragma unroll
        for (int mb = 0; mb < NB; ++mb)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
                acc[mb][nb] = Mma<DT>::run(Wf[(D1I_DBG & 2) ? 0 : kh % G::WRING][nb], Pf[(D1I_DBG & 1) ? 0 : (kh & 1)][mb], acc[mb][nb]);
        // issue order: four MFMAs, one fragment read; a weight load behind every third read. A K-half that writes a slice: the SPT writes one by
        // one, then the NB reads over the remaining gaps; the activation loads one by one from the first gap on.
        d3i_unroll([&](auto MBc) __attribute__((always_inline)) {
            constexpr int mb = decltype(MBc)::value;
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            if constexpr (!wr) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            } else if constexpr (mb < G::SPT) {
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            } else {
                constexpr int RG = NB - G::SPT;
                __builtin_amdgcn_sched_group_barrier(0x100, (NB * (mb - G::SPT + 1)) / RG - (NB * (mb - G::SPT)) / RG, 0);
            }
            if constexpr ((mb + 1) * 4 / NB != mb * 4 / NB) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            if constexpr (ld && mb >= NB - G::SPT) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }, std::make_integer_sequence<int, NB>{});
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (first && s + 1 < G::SLICES && s >= 1 && (D1I_DBG & 8) == 0) {
            // the slice's barrier: this wave's writes of slice s + 1 (a slice old) are done - behind them it has issued more LDS operations than
            // the counter holds; nobody reads slice s + 1 before everybody's pieces of it are in LDS, nobody overwrites column (s + 3) % 4 =
            // slice s - 1 (next slice's writes) before everybody has left slice s - 1
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NB) : "memory");
            d3q_sync();
        }
    };
    d3i_unroll([&](auto KHc) __attribute__((always_inline)) { step(KHc); }, std::make_integer_sequence<int, G::KH>{});
#ifdef D1I_CYCLES
    cyc__[2] = __builtin_amdgcn_s_memtime();
#endif

    // ---- epilogue: BN, activation, skip tensor, 16-byte stores (d3i_conv.hpp) ----
    const float alo = (p.act == PCV_ACT_RELU || p.act == PCV_ACT_RELU6) ? 0.f : -INFINITY, ahi = p.act == PCV_ACT_RELU6 ? 6.f : INFINITY;
    const float plo = (p.post_act == PCV_ACT_RELU || p.post_act == PCV_ACT_RELU6) ? 0.f : -INFINITY, phi = p.post_act == PCV_ACT_RELU6 ? 6.f : INFINITY;
    const float clo = alo > plo ? alo : plo, chi = ahi < phi ? ahi : phi;
    F16Guard<DT> guard;
    u32x4 rr[2][NB];
    if (has_res) {
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int mb = 0; mb < NB; ++mb) {
                const int m = m0 + 16 * mb + fr, ch0 = chw + 32 * g;
                const uint32_t roff = (ch0 < p.Cout && m < p.M) ? (uint32_t)((m * p.Cout + ch0) * 2) : 0x80000000u;
                rr[g][mb] = __builtin_amdgcn_raw_buffer_load_b128(rrsrc, roff, 0, 0);
            }
    }
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
            const int m = m0 + 16 * mb + fr;
            const uint32_t boff = (chok && m < p.M) ? (uint32_t)((m * p.Ypitch + ch0) * 2) : 0x80000000u;     // (the host keeps y below 2 GiB)
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
#ifdef D1I_CYCLES
    __builtin_amdgcn_s_waitcnt(0x0070);
    cyc__[3] = __builtin_amdgcn_s_memtime();
    if (p.dbg != nullptr && lane == 0) {
        uint32_t* d = p.dbg + (blockIdx.x * 4 + wave) * 8;
        for (int i = 0; i < 3; ++i) d[i] = (uint32_t)(cyc__[i + 1] - cyc__[i]);
        d[5] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - rt0__); d[6] = (uint32_t)rt0__; d[7] = 1u;
    }
#endif
}
#endif  // __HIP_DEVICE_COMPILE__

template <int DT, int CIN>
__global__ __launch_bounds__(256, 1) void d1i_kernel(const D3Params p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    d1i_body<DT, CIN>(p, smem);
#endif  // __HIP_DEVICE_COMPILE__
}

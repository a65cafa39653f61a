// p1r_conv.hpp - 1x1 convolution (stride 1 or 2) with 256 or 512 INPUT channels, gfx950 MFMA: the weights of a channel group sit in
// the REGISTERS of the eight waves, only the activations travel through LDS, and a pixel block's epilogue runs under the next
// block's MFMAs.
//
// Replaces: nn.Conv2d(Cin -> Cout, 1x1, stride s) + nn.BatchNorm2d(eval) + activation of ConvBlock.forward (reference
//           pytorchcv/models/common/conv.py:278-286) at ResBottleneck.conv3 / ResNeXtBottleneck.conv3 (resnet.py:128-131,
//           resnext.py:63-66) with the residual add + ReLU of the unit (resnet.py:227-228, resnext.py:131-132) in the epilogue, and
//           the strided identity convolution of ResUnit / ResNeXtUnit (resnet.py:200-206, resnext.py:107-113). Same K order
//           (channel slices in order), same MFMA chain per accumulator, same epilogue arithmetic as d3q_conv.hpp's 1x1 mode and
//           igemm_conv.hpp: bit-identical results.
//
// Why (round 4). The general 1x1 kernels stream BOTH operands through LDS: per K-step a 256 x 224 tile pulls 60 KB for 1 792 MFMA
// cycles (34 B/clk per CU where ~16 arrive next to a running matrix pipe, profiles/experiments/r04_d3w_loop.md), and with K = 256
// or 512 a tile is only 4-8 K-steps long, so its epilogue (up to 51 of 110 us on the 256 -> 512 stride-2 layer) is never hidden.
// Here:
//   * a block owns a channel GROUP of 8 x CW channels (CW = 64 with 256 input channels, 32 with 512; a third form, 32 x 256, for the
//     256-channel layers with a skip tensor or fewer than 384 output channels): wave w holds the CW x Cin weights of its channels as MFMA A
//     fragments - 128 (64) registers per lane, two waves per SIMD - loaded once per run of tiles;
//   * a tile is 64 KB of activations (128 or 64 pixels, all input channels) staged ONCE by LDS-DMA: 8-16 B/clk per CU; every
//     wave reads every fragment (one ds_read_b128 per 4 or 2 MFMAs);
//   * pixel-block-outer K loop (d3c_conv.hpp): a 16-pixel unit runs through all K-halves, then its BN / activation / residual /
//     store parts are written between the MFMAs of the next unit; two tile slots, ONE barrier per tile;
//   * the partial last round of a persistent grid is split into its 16-pixel units, one per block (`tailN`, below).
// LDS image of a unit (16 pixels): [64-channel slice][16 rows x 128 B], 16-byte chunk slot s of row R holds K-chunk s ^ (R & 7).
#pragma once
#include <type_traits>
#include "pcv_common.hpp"
#include "igemm_conv.hpp"     // Mma<DT>
#include "d3q_conv.hpp"       // D3Params, D3Tiles, d3q_tiles

template <int CW_, int CIN_>
struct P1RCfg {
    static constexpr int CW = CW_;                           // channels per wave
    static constexpr int CIN = CIN_;
    static constexpr int CWB = CW / 16;                      // 16-row MFMA blocks per wave
    static constexpr int KH = CIN / 32;                      // K-halves (one MFMA deep each)
    static constexpr int AREGS = CWB * KH * 4;               // registers per lane that hold the wave's weights: 128 (32 x 256: 64)
    static constexpr int NW = 8;                             // waves, two per SIMD (256 registers each). (Four waves of 64 channels x 512 with
                                                             // 256 weight registers, one per SIMD, half the LDS fragment reads: measured twice
                                                             // and dropped. With the compiler's MFMAs it parks half of the weights in AGPRs and
                                                             // copies them back per use: 84 us where this form takes 75. With inline-asm MFMAs
                                                             // whose A operand is constrained to "a" - 256 AGPRs of weights, no copy, no spill,
                                                             // bit-identical - 85 against 78, 48.5 against 44.6 at 7x7: a wave alone on its SIMD
                                                             // has no partner to cover its epilogue and memory waits.)
    static constexpr int THREADS = 64 * NW;
    static constexpr int CG = NW * CW;                       // channels per block: 512 / 256 / 256
    static constexpr int BM = CG;
    static constexpr int UNITB = 16 * CIN * 2;               // one 16-pixel unit: 8 / 16 KB
    static constexpr int SLOT = 65536;
    static constexpr int TP = SLOT / UNITB;                  // units per tile: 8 / 4
    static constexpr int BP = 16 * TP;                       // pixels per tile: 128 / 64
    static constexpr int PPU = CIN / 32;                     // 1 KB pieces per unit
    static constexpr int NPW = 64 / NW;                      // pieces per wave per tile (64 pieces)
    static constexpr int SSOFF = 2 * SLOT;                   // fp32 scale[CG] | shift[CG]
    static constexpr int LDS = SSOFF + 2 * CG * 4;
    static constexpr int NSTEP = TP * KH;                    // 64
    static constexpr int NSTORE = TP * (CWB / 2);            // 16-byte stores per lane per tile (all issued behind the tile's DMA pieces)
    static constexpr bool RES = CW == 32;                    // a skip tensor's pieces of a whole tile fit the registers (4 / 8 x 16 bytes per lane)
    static_assert((CW == 64 && CIN == 256) || (CW == 32 && CIN == 512) || (CW == 32 && CIN == 256), "three configurations");
    static_assert(NSTEP == 64 && TP * PPU == 64 && PPU % NW == 0 && KH % NPW == 0, "tile geometry");
};

#if defined(__HIP_DEVICE_COMPILE__)
template <int DT, int CW, int CIN>
__device__ __forceinline__ void p1r_body(const D3Params& p, char* smem) {
    typedef P1RCfg<CW, CIN> G;
    typedef typename Mma<DT>::frag frag;
    typedef __attribute__((address_space(3))) char lds_char;
    typedef const __attribute__((address_space(3))) frag* lds_fptr;
    constexpr int KH = G::KH, CWB = G::CWB, TP = G::TP;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int lrow = lane >> 3;
    const uint32_t lds0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(size_t)PCV_LDS(smem));
    const D3Tiles T = d3q_tiles(p);
    if (T.nMine == 0) return;
#ifdef P1R_CYCLES
    const uint64_t cstart__ = __builtin_amdgcn_s_memtime();
#endif

    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    const bool has_res = p.res != nullptr;
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res), 0, has_res ? p.res_bytes : 0u, 0x00020000);

    // ---- weights: A fragments of this wave's CW channels, all K-halves (reloaded when the channel group changes) ----
    frag A[KH][CWB];
    const uint32_t ssaddr = lds0 + (uint32_t)(G::SSOFF + (wave * CW + 8 * fq) * 4);     // this lane's 8 channels of the first 32-channel group
    auto load_weights = [&](int cg) __attribute__((always_inline)) {
#pragma unroll
        for (int kh = 0; kh < KH; ++kh)
#pragma unroll
            for (int i = 0; i < CWB; ++i) {
                const uint32_t row = (uint32_t)(cg * G::CG + wave * CW + i * 16 + fr);
                const uint32_t off = (row * (uint32_t)p.Kpad + (uint32_t)((kh >> 1) * 64 + (fq + 4 * (kh & 1)) * 8)) * 2u;     // rows past the blob: zeros
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, off, 0, 0);
                A[kh][i] = __builtin_bit_cast(frag, v);
            }
        // BN constants of the wave's channels into its own LDS strip (read back by this wave only: no barrier)
        if (lane < CW) {
            const int ch = cg * G::CG + wave * CW + lane;
            const float sc = ch < p.Cout ? p.scale[ch] : 0.f, sh = ch < p.Cout ? p.shift[ch] : 0.f;
            *reinterpret_cast<__attribute__((address_space(3))) float*>((size_t)(lds0 + (uint32_t)(G::SSOFF + (wave * CW + lane) * 4))) = sc;
            *reinterpret_cast<__attribute__((address_space(3))) float*>((size_t)(lds0 + (uint32_t)(G::SSOFF + (G::CG + wave * CW + lane) * 4))) = sh;
        }
    };

    // ---- fragment address of row fr, K-half parity 0 (parity 1: bit 6 flipped); + slot, + unit * UNITB + slice * 2048 as immediates ----
    const uint32_t fbase = lds0 + (uint32_t)(fr * 128 + ((fq ^ (fr & 7)) << 4));

    // ---- DMA: piece idx = wave + NW i of the tile's 64; PPU pieces per unit: (slice, row half) = idx % PPU ----
    const int half8 = wave & 1;
    const uint32_t lanesrc = (uint32_t)((((lane & 7) ^ lrow) << 4));
    int tileP0N = 0;                                              // first output pixel of the NEXT tile (or of the first, in the prologue)
    bool moreN = false;
    bool pseudoN = false;                                         // the next "tile" is one 16-pixel unit of a split tail tile: only its unit 0 exists
    uint32_t pixoff = 0x80000000u;                                // byte offset of this lane's input pixel of the unit being staged
    auto unit_src = [&](int unit) __attribute__((always_inline)) {
        const int m = tileP0N + unit * 16 + half8 * 8 + lrow;
        uint32_t mi = (uint32_t)m;
        if (p.stride != 1) {
            const uint32_t n = fastdiv((uint32_t)m, p.div_hw);
            const uint32_t rem = (uint32_t)m - n * (uint32_t)p.HW;
            const uint32_t ho = fastdiv(rem, p.div_w);
            const uint32_t wo = rem - ho * (uint32_t)p.W;
            mi = (n * (uint32_t)p.Hin + ho * (uint32_t)p.stride) * (uint32_t)p.Win + wo * (uint32_t)p.stride;
        }
        pixoff = (moreN && m < p.M && !(pseudoN && unit > 0)) ? mi * (uint32_t)(G::CIN * 2) + lanesrc : 0x80000000u;       // (the host keeps x below 2 GiB)
    };
    auto dma_piece = [&](auto Ic, int slot) __attribute__((always_inline)) {
        constexpr int I = decltype(Ic)::value;
        constexpr int unit = (G::NW * I) / G::PPU, in0 = (G::NW * I) % G::PPU;      // piece wave + NW I: unit, (slice, row half) = in0 + wave
        constexpr int sl0 = in0 / 2;                               // slice = (wave >> 1) + sl0
        if constexpr (in0 == 0) unit_src(unit);
        const int slice = (wave >> 1) + sl0;
        const uint32_t dst = lds0 + (uint32_t)(slot * G::SLOT + unit * G::UNITB + slice * 2048 + half8 * 1024);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_char*)(size_t)dst, 16, pixoff, slice * 128, 0, 0);
    };

    constexpr int RD = 3, RING = 6, KEEP = RING - RD - 1, NSTEP = G::NSTEP;       // (a short ring: 256 registers per lane)
    f32x4 cacc[2][CWB];                                           // [unit parity][16-channel block]
    frag bq[RING];
    uint32_t fs0 = 0, fs1 = 0;                                    // fbase + slot, K-half parity 0 / 1
    auto rd = [&](auto STc) __attribute__((always_inline)) {
        constexpr int st = decltype(STc)::value;
        constexpr int u = st / KH, kh = st - KH * u, s = kh >> 1, h = kh & 1;
        bq[st % RING] = *reinterpret_cast<lds_fptr>((size_t)(h ? fs1 : fs0) + (size_t)(u * G::UNITB + s * 2048));
    };
    // activations as branch-free clamps to launch-uniform bounds (d3c_conv.hpp)
    const float alo = (p.act == PCV_ACT_RELU || p.act == PCV_ACT_RELU6) ? 0.f : -INFINITY, ahi = p.act == PCV_ACT_RELU6 ? 6.f : INFINITY;
    const float plo = (p.post_act == PCV_ACT_RELU || p.post_act == PCV_ACT_RELU6) ? 0.f : -INFINITY, phi = p.post_act == PCV_ACT_RELU6 ? 6.f : INFINITY;
    const float clo = alo > plo ? alo : plo, chi = ahi < phi ? ahi : phi;
    constexpr int NRR = G::RES ? TP * (CWB / 2) : 1;              // residual pieces of a tile (32 channels per wave only: 4 / 8)
    u32x4 rrq[NRR];
    u32x4 opend;
    // BN constants of (32-channel group ip, accumulator half): two buffers (half 0 / 1), read from the wave's LDS strip at least a step
    // ahead of the part that uses them - a read directly in front of its use waits with lgkmcnt(0), i.e. for every fragment read in flight
    // as well (measured: the epilogue then cost more than the K loop)
    f32x4 scA, shA, scB, shB;
    typedef const __attribute__((address_space(3))) f32x4* lds_f4ptr;
    auto ss_read = [&](auto IPc, auto Hc) __attribute__((always_inline)) {
        constexpr int ip = decltype(IPc)::value, half = decltype(Hc)::value;
        const f32x4 a = *reinterpret_cast<lds_f4ptr>((size_t)ssaddr + (size_t)((ip * 32 + 4 * half) * 4));
        const f32x4 b = *reinterpret_cast<lds_f4ptr>((size_t)ssaddr + (size_t)((G::CG + ip * 32 + 4 * half) * 4));
        if constexpr (half == 0) { scA = a; shA = b; } else { scB = a; shB = b; }
        __builtin_amdgcn_sched_barrier(0);                        // (left alone, the scheduler sinks the read to its first use)
    };
    // part P of unit U's epilogue: 32-channel group ip = P / 4, output dword e = P % 4 (values 2 e, 2 e + 1 of the lane's 8 channels)
    // MODE of a launch's epilogue (uniform; one instantiation of the tile body each): bit 0 = skip tensor, bit 1 = an activation in FRONT
    // of the add, bit 2 = a finite upper bound (ReLU6) somewhere. The common forms - ReLU / none, the bottleneck's add + ReLU - then cost
    // 4 / 8 vector instructions per two values (packed fp32 FMA / add, one IEEE maximum each, one packed convert): with two waves per
    // SIMD the matrix pipe leaves ~3 issue slots per MFMA, and the first form of this epilogue (clamp to both bounds in every case,
    // address arithmetic per store) took as long as the K loop.
    auto epi_part = [&](auto Mc, auto Uc, auto Pc, F16Guard<DT>& guard) __attribute__((always_inline)) {
        constexpr int MODE = decltype(Mc)::value;
        constexpr bool HR = (MODE & 1) != 0, PRE = (MODE & 2) != 0, HI = (MODE & 4) != 0;
        constexpr int u = decltype(Uc)::value, P = decltype(Pc)::value;
        constexpr int ip = P >> 2, e = P & 3, half = e >> 1, k0 = 2 * (e & 1);
        const f32x4& scv = half == 0 ? scA : scB;                 // requested a step or more ahead (ss_read)
        const f32x4& shv = half == 0 ? shA : shB;
        const f32x4& c = cacc[u & 1][2 * ip + half];
        f32x2 v = (f32x2){c[k0], c[k0 + 1]} * (f32x2){scv[k0], scv[k0 + 1]} + (f32x2){shv[k0], shv[k0 + 1]};      // (contracted: one packed FMA)
        if constexpr (HR) {
            if constexpr (PRE) {
                v[0] = __builtin_elementwise_maximum(v[0], alo); v[1] = __builtin_elementwise_maximum(v[1], alo);
                if constexpr (HI) { v[0] = __builtin_elementwise_minimum(v[0], ahi); v[1] = __builtin_elementwise_minimum(v[1], ahi); }
            }
            float lo, hi;
            unpack2<DT>(rrq[(u * (CWB / 2) + ip) % NRR][e], lo, hi);
            v += (f32x2){lo, hi};
            v[0] = __builtin_elementwise_maximum(v[0], plo); v[1] = __builtin_elementwise_maximum(v[1], plo);
            if constexpr (HI) { v[0] = __builtin_elementwise_minimum(v[0], phi); v[1] = __builtin_elementwise_minimum(v[1], phi); }
        } else {
            // nothing between the two activations: one clamp to (max(alo, plo), min(ahi, phi)) (d3c_conv.hpp)
            v[0] = __builtin_elementwise_maximum(v[0], clo); v[1] = __builtin_elementwise_maximum(v[1], clo);
            if constexpr (HI) { v[0] = __builtin_elementwise_minimum(v[0], chi); v[1] = __builtin_elementwise_minimum(v[1], chi); }
        }
        guard.see2(v[0], v[1]);
        opend[e] = pack2<DT>(v[0], v[1]);
    };
    // Output / skip-tensor offsets: per tile ONE product per lane (pixel mTile + fr, the lane's channels of 32-channel group ip; 2^31 =
    // never in range for channels past Cout), per unit + 16 rows. A pixel past M lies past the end of the tensor: the descriptor's
    // range check drops it (y_bytes / res_bytes are exact).
    uint32_t yb[CWB / 2], rb[CWB / 2];
    auto epi_store = [&](int u, int ip) __attribute__((always_inline)) {
        __builtin_amdgcn_raw_buffer_store_b128(opend, yrsrc, yb[ip] + (uint32_t)(u * 32 * p.Ypitch), 0, 0);      // (nt / sc1 stores measured: no difference)
    };
    // one tile: 64 steps (unit u, K-half kh); the wave's 8 pieces of the NEXT tile go out during unit 0 (into the other slot: every
    // wave left it before the barrier that ended the last tile), so that every store of this tile is issued behind them
    // ONE: a single-unit pseudo tile - unit 0's steps, then its epilogue on its own (its own instantiation: as a run-time branch inside the
    // full tile's code it cost that code 20 registers it does not have)
    auto tile_steps = [&](auto HRc, auto ONEc, int slot, int ch0, int mTile) __attribute__((always_inline)) {
        constexpr bool ONE = decltype(ONEc)::value;
        constexpr bool HR = (decltype(HRc)::value & 1) != 0;
        F16Guard<DT> guard;
#pragma unroll
        for (int ip = 0; ip < CWB / 2; ++ip) {
            const bool chok = ch0 + 32 * ip < p.Cout;
            yb[ip] = chok ? (uint32_t)(((mTile + fr) * p.Ypitch + ch0 + 32 * ip) * 2) : 0x80000000u;      // (the host keeps y below 2 GiB)
            if constexpr (HR) rb[ip] = chok ? (uint32_t)(((mTile + fr) * p.Cout + ch0 + 32 * ip) * 2) : 0x80000000u;
        }
        if constexpr (HR) {
            // the skip tensor's pieces of the whole tile, in FRONT of the LDS-DMA pieces (vmcnt retires in order)
#pragma unroll
            for (int u = 0; u < (ONE ? 1 : TP); ++u)
#pragma unroll
                for (int ip = 0; ip < CWB / 2; ++ip) {
                    rrq[(u * (CWB / 2) + ip) % NRR] = __builtin_amdgcn_raw_buffer_load_b128(rrsrc, rb[ip] + (uint32_t)(u * 32 * p.Cout), 0, 0);
                }
        }
        rd(std::integral_constant<int, 0>{}); rd(std::integral_constant<int, 1>{}); rd(std::integral_constant<int, 2>{});
        static_assert(RD == 3, "the three reads above");
        auto step = [&](auto STc) __attribute__((always_inline)) {
            constexpr int st = decltype(STc)::value;
            constexpr int u = st / KH, kh = st - KH * u;
            if constexpr (st + RD < NSTEP) rd(std::integral_constant<int, st + RD>{});
            if constexpr (u == 0 && (kh % (KH / G::NPW)) == 0) dma_piece(std::integral_constant<int, kh / (KH / G::NPW)>{}, slot ^ 1);
#pragma unroll
            for (int i = 0; i < CWB; ++i) {
                if constexpr (kh == 0) cacc[u & 1][i] = Mma<DT>::run(A[0][i], bq[st % RING], (f32x4){0.f, 0.f, 0.f, 0.f});
                else cacc[u & 1][i] = Mma<DT>::run(A[kh][i], bq[st % RING], cacc[u & 1][i]);
            }
            // (the fragment of two steps ago stays alive up to here: d3c_conv.hpp, write-after-read on an MFMA source operand)
            // (anchored behind this step's MFMAs through their accumulator, as a "v" operand: an "a" operand makes the compiler split the
            // 256 registers into 128 + 128 AGPRs)
            if constexpr (st >= KEEP) asm volatile("" ::"v"(bq[(st - KEEP) % RING]), "v"(cacc[u & 1][CWB - 1]));
            if constexpr (u >= 1) {                                 // unit u - 1 is finished under this unit's MFMAs
                typedef std::integral_constant<int, u - 1> UP;
                typedef std::integral_constant<int, 0> I0;
                typedef std::integral_constant<int, 1> I1;
                auto part = [&](auto Pc) __attribute__((always_inline)) { epi_part(HRc, UP{}, Pc, guard); };
                typedef std::integral_constant<int, 2> I2; typedef std::integral_constant<int, 3> I3; typedef std::integral_constant<int, 4> I4;
                typedef std::integral_constant<int, 5> I5; typedef std::integral_constant<int, 6> I6; typedef std::integral_constant<int, 7> I7;
                if constexpr (CWB == 4 && KH == 8) {                // 8 parts + 2 stores over 8 steps
                    if constexpr (kh == 0) { ss_read(I0{}, I0{}); ss_read(I0{}, I1{}); }
                    if constexpr (kh == 2) { part(I0{}); part(I1{}); ss_read(I1{}, I0{}); }
                    if constexpr (kh == 3) { part(I2{}); part(I3{}); ss_read(I1{}, I1{}); epi_store(u - 1, 0); }
                    if constexpr (kh == 5) { part(I4{}); part(I5{}); }
                    if constexpr (kh == 6) { part(I6{}); part(I7{}); }
                    if constexpr (kh == 7) epi_store(u - 1, 1);
                } else if constexpr (KH == 8) {                     // 4 parts + 1 store over 8 steps
                    if constexpr (kh == 0) { ss_read(I0{}, I0{}); ss_read(I0{}, I1{}); }
                    if constexpr (kh == 2) part(I0{});
                    if constexpr (kh == 3) part(I1{});
                    if constexpr (kh == 5) part(I2{});
                    if constexpr (kh == 6) part(I3{});
                    if constexpr (kh == 7) epi_store(u - 1, 0);
                } else {                                            // 4 parts + 1 store over 16 steps
                    if constexpr (kh == 0) { ss_read(I0{}, I0{}); ss_read(I0{}, I1{}); }
                    if constexpr (kh == 3) part(I0{});
                    if constexpr (kh == 5) part(I1{});
                    if constexpr (kh == 8) part(I2{});
                    if constexpr (kh == 10) part(I3{});
                    if constexpr (kh == 12) epi_store(u - 1, 0);
                }
            }
        };
        auto run = [&](auto... Sc) __attribute__((always_inline)) { (step(Sc), ...); };
        auto eight = [&](auto Bc) __attribute__((always_inline)) {       // steps 8 B .. 8 B + 7
            constexpr int B0 = decltype(Bc)::value * 8;
            run(std::integral_constant<int, B0>{}, std::integral_constant<int, B0 + 1>{}, std::integral_constant<int, B0 + 2>{},
                std::integral_constant<int, B0 + 3>{}, std::integral_constant<int, B0 + 4>{}, std::integral_constant<int, B0 + 5>{},
                std::integral_constant<int, B0 + 6>{}, std::integral_constant<int, B0 + 7>{});
        };
        auto last_epilogue = [&](auto ULc) __attribute__((always_inline)) {       // a unit's epilogue on its own (nothing left to hide it under)
            typedef decltype(ULc) UL;
            constexpr int ul = UL::value;
            ss_read(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}); ss_read(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
            epi_part(HRc, UL{}, std::integral_constant<int, 0>{}, guard); epi_part(HRc, UL{}, std::integral_constant<int, 1>{}, guard);
            epi_part(HRc, UL{}, std::integral_constant<int, 2>{}, guard); epi_part(HRc, UL{}, std::integral_constant<int, 3>{}, guard);
            epi_store(ul, 0);
            if constexpr (CWB == 4) {
                ss_read(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}); ss_read(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
                epi_part(HRc, UL{}, std::integral_constant<int, 4>{}, guard); epi_part(HRc, UL{}, std::integral_constant<int, 5>{}, guard);
                epi_part(HRc, UL{}, std::integral_constant<int, 6>{}, guard); epi_part(HRc, UL{}, std::integral_constant<int, 7>{}, guard);
                epi_store(ul, 1);
            }
        };
        eight(std::integral_constant<int, 0>{});
        if constexpr (KH == 16) eight(std::integral_constant<int, 1>{});
        if constexpr (ONE) last_epilogue(std::integral_constant<int, 0>{});
        else {
            if constexpr (KH == 8) eight(std::integral_constant<int, 1>{});
            eight(std::integral_constant<int, 2>{});
            eight(std::integral_constant<int, 3>{}); eight(std::integral_constant<int, 4>{}); eight(std::integral_constant<int, 5>{});
            eight(std::integral_constant<int, 6>{}); eight(std::integral_constant<int, 7>{});
            last_epilogue(std::integral_constant<int, TP - 1>{});
        }
        guard.commit(p.ovf);
    };

    // ---- prologue: the first tile's activations ----
    int cg = T.tile0 % p.nChTiles;
    int tileP0 = (T.tile0 / p.nChTiles) * G::BP;
    tileP0N = tileP0; moreN = true;
    {
        auto all = [&](auto... Ic) __attribute__((always_inline)) { (dma_piece(Ic, 0), ...); };
        auto eight = [&](auto Bc) __attribute__((always_inline)) {
            constexpr int B0 = decltype(Bc)::value;
            all(std::integral_constant<int, B0>{}, std::integral_constant<int, B0 + 1>{}, std::integral_constant<int, B0 + 2>{},
                std::integral_constant<int, B0 + 3>{}, std::integral_constant<int, B0 + 4>{}, std::integral_constant<int, B0 + 5>{},
                std::integral_constant<int, B0 + 6>{}, std::integral_constant<int, B0 + 7>{});
        };
        eight(std::integral_constant<int, 0>{});
    }

    // The tail: when the tile count is not a multiple of the grid, the host splits the last (partial) round's tiles into their 16-pixel
    // units, one per block - a block's unit belongs to a tile of ITS channel group (block j of an XCD owns group j % nCG; the r-th block
    // of a group takes unit r % TP of the group's tail tile r / TP). It runs as a one-unit pseudo tile behind the block's last tile.
    int tailP0 = -1;
    if (p.tailN > 0) {
        const int nCG = p.nChTiles, B = gridDim.x >> 3, j = blockIdx.x >> 3, xcd = blockIdx.x & 7;
        const int r = xcd * (B / nCG) + j / nCG, k = r / TP, u = r - k * TP;
        if (k < p.tailN / nCG) tailP0 = (p.nTiles / nCG + k) * G::BP + 16 * u;
    }
    bool pseudo = false;
    int slot = 0, t = T.tile0;
    // a RUN of tiles that share their channel group: weights and BN constants are loaded in front of the run (d3c_conv.hpp)
    while (t < T.tend) {
#ifdef P1R_CYCLES
        const uint64_t p0__ = __builtin_amdgcn_s_memtime();
#endif
        load_weights(cg);
#ifdef P1R_CYCLES
        const uint64_t p1__ = __builtin_amdgcn_s_memtime();
#endif
        __builtin_amdgcn_s_waitcnt(0x0070);                        // vmcnt(0) lgkmcnt(0), visible to the compiler's wait-count pass
#ifdef P1R_CYCLES
        const uint64_t p2__ = __builtin_amdgcn_s_memtime();
#endif
        d3q_sync();
#ifdef P1R_CYCLES
        if (p.dbg != nullptr && t == T.tile0 && lane == 0 && wave == 0) {
            uint32_t* e = p.dbg + 256 * 8 * 4 + blockIdx.x * 32 + 30;
            e[0] = (uint32_t)(p0__ - cstart__) | ((uint32_t)(p1__ - p0__) << 16);
            e[1] = (uint32_t)(p2__ - p1__) | ((uint32_t)(__builtin_amdgcn_s_memtime() - p2__) << 16);
        }
#endif
        bool same;
        do {
            const int tn = t + T.tstride;
            moreN = tn < T.tend;
            int cgN = moreN ? tn % p.nChTiles : cg;
            tileP0N = moreN ? (tn / p.nChTiles) * G::BP : 0;
            pseudoN = !moreN && !pseudo && tailP0 >= 0;          // behind the last tile: this block's unit of the split tail
            if (pseudoN) { moreN = true; cgN = cg; tileP0N = tailP0; }
            fs0 = fbase + (uint32_t)(slot * G::SLOT);
            fs1 = fs0 ^ 64u;
#ifdef P1R_CYCLES      // diagnostic build (tests/tools/p1r_cycles.py): shader-cycle stamps of this block's third tile
            const bool stamp__ = p.dbg != nullptr;
            const int tl__ = (t - T.tile0) / T.tstride;
            uint64_t c0__ = 0, c1__ = 0, c2__ = 0;
            if (stamp__) c0__ = __builtin_amdgcn_s_memtime();
#endif
            {
                const int ch0 = cg * G::CG + wave * CW + 8 * fq;
                // (uniform branches; MODE bits: epi_part)
                const bool hi = p.act == PCV_ACT_RELU6 || p.post_act == PCV_ACT_RELU6, pre = p.act != PCV_ACT_NONE;
                auto modes = [&](auto ONEc) __attribute__((always_inline)) {
                    if constexpr (G::RES) {
                        if (has_res) {
                            if (!hi && !pre) tile_steps(std::integral_constant<int, 1>{}, ONEc, slot, ch0, tileP0);
                            else tile_steps(std::integral_constant<int, 7>{}, ONEc, slot, ch0, tileP0);
                        } else if (!hi) tile_steps(std::integral_constant<int, 0>{}, ONEc, slot, ch0, tileP0);
                        else tile_steps(std::integral_constant<int, 4>{}, ONEc, slot, ch0, tileP0);
                    } else {
                        if (!hi) tile_steps(std::integral_constant<int, 0>{}, ONEc, slot, ch0, tileP0);
                        else tile_steps(std::integral_constant<int, 4>{}, ONEc, slot, ch0, tileP0);
                    }
                };
                if (pseudo) modes(std::true_type{});
                else modes(std::false_type{});
            }
#ifdef P1R_CYCLES
            if (stamp__) c1__ = __builtin_amdgcn_s_memtime();
#endif
            // the next tile has landed (behind its pieces only this tile's NSTORE stores were issued); every wave is done with this one
            if constexpr (G::NSTORE == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if constexpr (G::NSTORE == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
#ifdef P1R_CYCLES
            if (stamp__) c2__ = __builtin_amdgcn_s_memtime();
#endif
            d3q_sync();
#ifdef P1R_CYCLES
            if (stamp__ && lane == 0) {
                const uint64_t c3 = __builtin_amdgcn_s_memtime();
                if (tl__ == 2) {
                    uint32_t* d = p.dbg + (blockIdx.x * 8 + wave) * 4;      // (8 records per block whatever the wave count)
                    d[0] = (uint32_t)(c1__ - c0__); d[1] = (uint32_t)(c2__ - c1__); d[2] = (uint32_t)(c3 - c2__); d[3] = 1u;
                }
                if (wave == 0 && tl__ < 15) {                       // per tile: start (relative to the kernel's first instruction) and duration
                    uint32_t* e = p.dbg + 256 * 8 * 4 + blockIdx.x * 32 + 2 * tl__;
                    e[0] = (uint32_t)(c0__ - cstart__); e[1] = (uint32_t)(c3 - c0__);
                }
            }
#endif
            same = moreN && cgN == cg;
            cg = cgN; tileP0 = tileP0N;
            pseudo = pseudoN;
            slot ^= 1;
            t = tn;
        } while (same);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // (a pseudo tile's out-of-range DMA pieces: nothing may land in LDS after the block has left)
}
#endif  // __HIP_DEVICE_COMPILE__

template <int DT, int CW, int CIN>
__global__ __launch_bounds__((P1RCfg<CW, CIN>::THREADS), 1) void p1r_kernel(const D3Params p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    p1r_body<DT, CW, CIN>(p, smem);
#endif  // __HIP_DEVICE_COMPILE__
}

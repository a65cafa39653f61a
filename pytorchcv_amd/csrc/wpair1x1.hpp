// wpair1x1.hpp - the fused 1x1 pair of pair1x1.hpp for the WIDER bottleneck stages: the last convolution of a unit
// (CM -> 4 CM, BN, + identity, ReLU; reference resnet.py:227-228) and the first convolution of the next unit
// (4 CM -> CM, BN, ReLU; resnet.py:106-109), CM = 128 / 256 (ResNet-50/101/152 stages 2 and 3), 16-bit storage. The same with
// C1 = 2 CM covers ResNeXt 32x4d stages 1 and 2 (128 -> 256 -> 128, 256 -> 512 -> 256; resnext.py:62-80).
//
// Both layers are HBM-bound when run separately (57 FLOP/B at CM = 128) and the second one re-reads the 4 CM-channel
// tensor the first has just written: 717 MB per pair at batch 256 against 512 MB when y1 goes straight from the first
// epilogue into the second GEMM. The weights (2 x 128 KB and more) no longer fit the register file as in pair1x1.hpp, so
// they stream through LDS, one 64-channel chunk of y1 at a time, while the pixels stay put:
//
//   CM = 128: block = 4 waves, tile = 128 pixels, wave w owns pixels 32w .. 32w+31 for BOTH GEMMs
//   x fragments of the wave's pixels: global -> registers once per tile (K = CM)
//   for chunk c of 64 y1 channels (4 CM / 64 chunks):
//       A1 = W1 rows [64c, 64c+64) x K=CM and A2 = W2 rows [0, CM) x K-slice [64c, 64c+64) in LDS
//       GEMM1  acc1 = A1 . x
//       epi 1  scale/shift, + residual, ReLU, round; 16-byte NHWC stores of y1; the packs ARE GEMM2's B fragments
//       GEMM2  acc2 += A2 . y1chunk
//   epi 2  scale/shift, ReLU, 16-byte stores of y2
//   Two whole-chunk LDS slots, one barrier per chunk: { wait own DMA(c) ; barrier ; issue DMA(c+1) into the slot read in
//   chunk c-1 ; prefetch the residual of chunk c+1 ; compute }. 64 KB ring + 5 KB BN tables, 248 registers: 2 blocks per CU.
//
//   CM = 256: acc2 for 256 output channels x 32 pixels would not fit 256 registers beside the x fragments, so a PAIR of
//   waves shares 32 pixels: wave (wp, wc) computes the 32 channels ip = wc of the chunk in GEMM1, the two halves are swapped
//   through LDS (a second barrier in the chunk), and each wave accumulates its half of the 256 output channels in GEMM2.
//   A chunk's A1 + A2 are 64 KB there; the two-slot ring takes 128 KB, so the block has 8 waves (4 pixel groups x 2, tile =
//   128 pixels) and owns the CU alone: same 8 waves per CU, and the weights get a whole chunk of time to arrive. (Tried
//   first and rejected: one LDS buffer per operand with two 4-wave blocks per CU - every half-chunk then waits ~1.5 us for
//   a DMA issued only half a chunk earlier, 109 us against 102 us unfused; a wave owning 16 pixels alone did no better.)
#pragma once
#include "pcv_common.hpp"
#include "igemm_conv.hpp"     // Mma<DT>

struct WPairParams {
    const void* x;            // [M][CM]
    const void* w1;           // packed rows [4 CM][CM]  (MFMA row order)
    const void* res;          // [M][4 CM]
    void* y1;                 // [M][4 CM]
    const void* w2;           // packed rows [CM][4 CM]
    void* y2;                 // [M][CM]
    const float* scale1;
    const float* shift1;
    const float* scale2;
    const float* shift2;
    uint32_t x_bytes, res_bytes, y1_bytes, y2_bytes, w1_bytes, w2_bytes;
    int M, nTiles;
    int act1, post1, act2;
    // GATE: y1 = post1(act1(BN(acc)) * gate[n, c] + res), gate fp32 [N][C1] (an SE block run inside the first convolution,
    // pcv_conv2d_gated_fused's epilogue); n = pixel / HW
    const float* gate;
    FastDiv div_hw;
    uint32_t hw;
    uint32_t* ovf;            // the context's fp16 overflow counter (pcv_common.hpp, F16Guard)
};

template <int N> __device__ __forceinline__ void wpair_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int CM, int C1_ = 4 * CM> struct WPairCfg {
    static constexpr int C1 = C1_;                      // 4 CM in ResNet, 2 CM in ResNeXt 32x4d (resnext.py:62-80)
    static constexpr bool PAIRW = CM > 128;             // pairs of waves share their pixels and split the channels
    static constexpr int NWC = PAIRW ? 2 : 1;           // waves along the channels
    static constexpr int NWP = 4;                       // waves along the pixels
    static constexpr int NW = NWC * NWP;                // waves per block
    static constexpr int PBW = 2;                       // 16-pixel blocks per wave
    static constexpr int P = 16 * PBW * NWP;            // pixels per tile
    static constexpr int NCH = C1 / 64;                 // chunks of y1 channels
    static constexpr int KS1 = CM / 32;                 // MFMA K-steps of GEMM1
    static constexpr int NI2 = CM / 16;                 // 16-row fragments of GEMM2's output
    static constexpr int A1B = (CM / 64) * 64 * 128;    // bytes of A1 per chunk (CM / 64 slabs of 64 rows x 128 bytes)
    static constexpr int A2B = CM * 128;                // bytes of A2 per chunk
    static constexpr int SLOT = A1B + A2B;
    static constexpr int XCH = PAIRW ? NWP * 2 * PBW * 1024 : 0;      // exchange buffer [wp][ip][j][lane] x 16 bytes
    static constexpr int TAB = 2 * SLOT + XCH;                        // BN tables: sc1, sf1 [C1], sc2, sf2 [CM]
    static constexpr int LDS = TAB + (2 * C1 + 2 * CM) * 4;
    // GATE: the gate rows of the (at most two) images a tile touches, double buffered by tile parity - when there is room
    static constexpr bool GATE_LDS = LDS + 4 * C1 * 4 <= 80 * 1024;
    static constexpr int LDS_GATED = LDS + (GATE_LDS ? 4 * C1 * 4 : 0);
};

template <int DT, int CM, int C1_ = 4 * CM, bool GATE = false>
__global__ __launch_bounds__((64 * WPairCfg<CM, C1_>::NW), 2) void wpair1x1_kernel(const WPairParams p) {   // 2 waves per SIMD: 256 registers
#if defined(__HIP_DEVICE_COMPILE__)
    typedef WPairCfg<CM, C1_> G;
    constexpr int C1 = G::C1, P = G::P, NCH = G::NCH, KS1 = G::KS1, PBW = G::PBW, NWC = G::NWC;
    constexpr bool PAIRW = G::PAIRW;
    constexpr int NW = G::NW, NT = 64 * NW;
    constexpr int NI1 = 4 / NWC;                        // GEMM1 row fragments per wave (of the chunk's 4)
    constexpr int NIP = 2 / NWC;                        // 32-channel groups of the chunk finished per wave
    constexpr int NI2W = G::NI2 / NWC;                  // GEMM2 row fragments per wave
    constexpr int YOUNG = 2 * NIP * PBW;                // residual loads + y1 stores a wave issues per chunk
    static_assert(NCH % 2 == 0, "the ring parity must be the same at every tile start");
    static_assert((G::A1B / 1024) % NW == 0 && (G::A2B / 1024) % NW == 0, "pieces must split evenly over the waves");
    constexpr int W1P = G::A1B / 1024 / NW, W2P = G::A2B / 1024 / NW;    // pieces per wave
    typedef typename Mma<DT>::frag frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const xch = smem + G::TAB - G::XCH;
    float* const tsc1 = reinterpret_cast<float*>(smem + G::TAB);
    float* const tsf1 = tsc1 + C1;
    float* const tsc2 = tsf1 + C1;
    float* const tsf2 = tsc2 + CM;
    float* const tgate = tsf2 + CM;                     // GATE && G::GATE_LDS: [tile parity][image 0 / 1 of the tile][C1]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave / NWC, wc = wave % NWC;
    const int fr = lane & 15, fq = lane >> 4;
    const int lrow = lane >> 3;
    const int cs = (lane & 7) ^ lrow;                            // source-side swizzle of the DMA pieces

    int tile = blockIdx.x;
    const int tstride = gridDim.x;
    if (tile >= p.nTiles) return;

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res), 0, p.res_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t y1rsrc = __builtin_amdgcn_make_buffer_rsrc(p.y1, 0, p.y1_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t y2rsrc = __builtin_amdgcn_make_buffer_rsrc(p.y2, 0, p.y2_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w1rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w1), 0, p.w1_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w2rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w2), 0, p.w2_bytes, 0x00020000);

    // ---- BN tables -> LDS (once per persistent block) ---------------------------------------------------------------
    for (int i = tid; i < C1; i += NT) { tsc1[i] = p.scale1[i]; tsf1[i] = p.shift1[i]; }
    for (int i = tid; i < CM; i += NT) { tsc2[i] = p.scale2[i]; tsf2[i] = p.shift2[i]; }

    // ---- weight buffers -----------------------------------------------------------------------------------------------
    // piece pc of A1: slab s = pc / 8, rows 8 (pc % 8) .. +8 of the chunk; piece pc of A2: rows 8 pc .. +8 of W2.
    // LDS row = 128 bytes, 16-byte slot (lane & 7) of row (8 pc + lrow) receives source chunk cs.
    uint32_t w1off[W1P], w2off[W2P];
#pragma unroll
    for (int q = 0; q < W1P; ++q) {
        const int pc = W1P * wave + q, s = pc >> 3, row = 8 * (pc & 7) + lrow;
        w1off[q] = (uint32_t)((row * CM + s * 64 + cs * 8) * 2);                 // + chunk * 64 rows
    }
#pragma unroll
    for (int q = 0; q < W2P; ++q) {
        const int row = 8 * (W2P * wave + q) + lrow;
        w2off[q] = (uint32_t)((row * C1 + cs * 8) * 2);                          // + chunk * 64 K-elements
    }
    // two whole-chunk slots [A1 | A2]
    auto issue_w1 = [&](int c, int slot) {
        char* base = smem + slot * G::SLOT;
#pragma unroll
        for (int q = 0; q < W1P; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w1rsrc, PCV_LDS(base + (W1P * wave + q) * 1024), 16, w1off[q],
                                                     c * (64 * CM * 2), 0, 0);
    };
    auto issue_w2 = [&](int c, int slot) {
        char* base = smem + slot * G::SLOT + G::A1B;
#pragma unroll
        for (int q = 0; q < W2P; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w2rsrc, PCV_LDS(base + (W2P * wave + q) * 1024), 16, w2off[q], c * 128, 0, 0);
    };
    // A fragment (rows 16 i + fr of a 128-byte-row slab, MFMA K-step kk of its 64 K-elements)
    const int aswz0 = (fq ^ (fr & 7)) << 4, aswz1 = ((fq + 4) ^ (fr & 7)) << 4;
    const int arow = fr * 128;

    // ---- pixel-side addresses: this wave's pixels 16 PBW wp + 16 j + fr -------------------------------------------------------
    auto pix_of = [&](int t, int j) -> long { return (long)t * P + 16 * PBW * wp + 16 * j + fr; };
    auto load_x = [&](int t, frag (&xf)[KS1][PBW]) {
#pragma unroll
        for (int j = 0; j < PBW; ++j) {
            const long pix = pix_of(t, j);
            const bool ok = t < p.nTiles && pix < p.M;
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                const uint32_t off = ok ? (uint32_t)((pix * CM + 32 * ks + 8 * fq) * 2) : 0x80000000u;
                xf[ks][j] = __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, off, 0, 0));
            }
        }
    };
    // element (ipl, j) of chunk c: pixel 16 j + fr, channels 64 c + 32 (NIP wc + ipl) + 8 fq .. +8 of the 4 CM-channel tensors
    auto off_c1 = [&](int t, int c, int ipl, int j) -> uint32_t {
        const long pix = pix_of(t, j);
        return (t < p.nTiles && pix < p.M) ? (uint32_t)((pix * C1 + 64 * c + 32 * (NIP * wc + ipl) + 8 * fq) * 2) : 0x80000000u;
    };
    auto load_res = [&](int t, int c, u32x4 (&r)[NIP][PBW]) {
#pragma unroll
        for (int ipl = 0; ipl < NIP; ++ipl)
#pragma unroll
            for (int j = 0; j < PBW; ++j) r[ipl][j] = __builtin_amdgcn_raw_buffer_load_b128(rrsrc, off_c1(t, c, ipl, j), 0, 0);
    };
    const ActClamp act1 = make_act(p.act1), post1 = make_act(p.post1), act2 = make_act(p.act2);

    // ---- prologue -------------------------------------------------------------------------------------------------------
    frag xf[KS1][PBW];
    u32x4 resr[2][NIP][PBW];                   // [parity of the chunk][ipl][j]
    issue_w1(0, 0);
    issue_w2(0, 0);
    load_x(tile, xf);
    load_res(tile, 0, resr[0]);
    bool first = true;
    bool tile_parity = false;

    while (true) {
        f32x4 acc2[NI2W][PBW];
#pragma unroll
        for (int i = 0; i < NI2W; ++i)
#pragma unroll
            for (int j = 0; j < PBW; ++j) acc2[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

        int gate_n0 = 0;
        float* gbuf = tgate;
        if constexpr (GATE && G::GATE_LDS) {
            // the tile's gate rows -> LDS (visible after chunk 0's barrier; the other parity buffer may still be read by waves
            // that are finishing the previous tile). Host guarantees HW >= P: a tile touches at most two images.
            const long first = (long)tile * P;
            gate_n0 = (int)fastdiv((uint32_t)(first < p.M ? first : p.M - 1), p.div_hw);
            const int nimg = p.M / (int)p.hw;
            gbuf = tgate + (tile_parity ? 2 * C1 : 0);
            for (int i = tid * 4; i < 2 * C1; i += NT * 4) {
                const int img = i / C1, ch = i - img * C1;
                const int n = gate_n0 + img < nimg ? gate_n0 + img : nimg - 1;
                *reinterpret_cast<f32x4*>(gbuf + i) = *reinterpret_cast<const f32x4*>(p.gate + (size_t)n * C1 + ch);
            }
        }
        auto chunk = [&](int c, auto PAR) {
            constexpr int par = decltype(PAR)::value;            // c & 1: ring slot and residual register set
            const int cn = c + 1 < NCH ? c + 1 : 0;              // the weights do not depend on the tile: the last chunk
            const int tn = c + 1 < NCH ? tile : tile + tstride;  // requests chunk 0 again, for the next tile
            // DMA(c) of this wave landed: younger VMEM ops are the residual prefetch of chunk c and the y1 stores of
            // chunk c-1; at a tile start also the y2 stores and the x loads: wait for everything there.
            if (c == 0) {
                wpair_wait_vmcnt<0>();
                if constexpr (GATE && G::GATE_LDS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the gate rows are written
            } else wpair_wait_vmcnt<YOUNG>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            // next chunk's weights into the slot every wave has finished reading (chunk c-1)
            issue_w1(cn, par ^ 1);
            issue_w2(cn, par ^ 1);
            load_res(tn, cn, resr[par ^ 1]);
            const char* a1b = smem + par * G::SLOT + arow;
            const char* a2b = a1b + G::A1B;

            // ---- GEMM1: row fragments NI1 wc .. of the chunk ------------------------------------------------------------------
            f32x4 acc1[NI1][PBW];
#pragma unroll
            for (int i = 0; i < NI1; ++i)
#pragma unroll
                for (int j = 0; j < PBW; ++j) acc1[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                const int s = ks >> 1, swz = (ks & 1) ? aswz1 : aswz0;
                frag a[NI1];
#pragma unroll
                for (int i = 0; i < NI1; ++i)
                    a[i] = *reinterpret_cast<const frag*>(a1b + s * 8192 + (NI1 * wc + i) * 2048 + swz);
#pragma unroll
                for (int i = 0; i < NI1; ++i)
#pragma unroll
                    for (int j = 0; j < PBW; ++j) acc1[i][j] = Mma<DT>::run(a[i], xf[ks][j], acc1[i][j]);
            }

            // ---- epilogue 1: BN, + residual, activation, round; the packs are GEMM2's B fragments ---------------------------
            u32x4 o[NIP][PBW];
            F16Guard<DT> guard;
#pragma unroll
            for (int ipl = 0; ipl < NIP; ++ipl) {
                const int ch = 64 * c + 32 * (NIP * wc + ipl) + 8 * fq;
                const f32x4 s0 = *reinterpret_cast<const f32x4*>(tsc1 + ch), s1 = *reinterpret_cast<const f32x4*>(tsc1 + ch + 4);
                const f32x4 h0 = *reinterpret_cast<const f32x4*>(tsf1 + ch), h1 = *reinterpret_cast<const f32x4*>(tsf1 + ch + 4);
#pragma unroll
                for (int j = 0; j < PBW; ++j) {
                    float v[8], r8[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = acc1[2 * ipl][j][e] * s0[e] + h0[e];
                        v[4 + e] = acc1[2 * ipl + 1][j][e] * s1[e] + h1[e];
                    }
                    apply_act8(v, act1);
                    if constexpr (GATE) {
#pragma clang fp contract(off)      // as in igemm_conv.hpp: the product is rounded before the skip add
                        const long pix = pix_of(tile, j);
                        const uint32_t n = fastdiv((uint32_t)(pix < p.M ? pix : p.M - 1), p.div_hw);
                        const float* gp = G::GATE_LDS ? gbuf + ((int)n - gate_n0) * C1 + ch : p.gate + (size_t)n * C1 + ch;
                        const f32x4 g0 = *reinterpret_cast<const f32x4*>(gp), g1 = *reinterpret_cast<const f32x4*>(gp + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v[e] *= g0[e]; v[4 + e] *= g1[e]; }
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) unpack2<DT>(resr[par][ipl][j][e], r8[2 * e], r8[2 * e + 1]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += r8[e];
                    apply_act8(v, post1);
                    guard.see(v);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[ipl][j][e] = pack2<DT>(v[2 * e], v[2 * e + 1]);
                }
            }
            guard.commit(p.ovf);
#pragma unroll
            for (int ipl = 0; ipl < NIP; ++ipl)
#pragma unroll
                for (int j = 0; j < PBW; ++j)
                    __builtin_amdgcn_raw_buffer_store_b128(o[ipl][j], y1rsrc, off_c1(tile, c, ipl, j), 0, 0);

            // ---- GEMM2: K-steps 2c, 2c+1 of the second convolution, row fragments NI2W wc .. ---------------------------------
            u32x4 ob[2][PBW];                                    // B fragments of both K-steps
            if constexpr (PAIRW) {
                // the pair of waves that shares these pixels swaps its halves of the chunk: [wp][ip][j][lane]. The buffer is
                // free: the partner read it before it arrived at this chunk's first barrier.
#pragma unroll
                for (int j = 0; j < PBW; ++j)
                    *reinterpret_cast<u32x4*>(xch + (((wp * 2 + wc) * PBW + j) * 64 + lane) * 16) = o[0][j];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int j = 0; j < PBW; ++j)
                        ob[kk][j] = *reinterpret_cast<const u32x4*>(xch + (((wp * 2 + kk) * PBW + j) * 64 + lane) * 16);
            } else {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int j = 0; j < PBW; ++j) ob[kk][j] = o[kk % NIP][j];
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int swz = kk ? aswz1 : aswz0;
#pragma unroll
                for (int h = 0; h < NI2W / 4; ++h) {
                    frag a[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        a[i] = *reinterpret_cast<const frag*>(a2b + (NI2W * wc + 4 * h + i) * 2048 + swz);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < PBW; ++j)
                            acc2[4 * h + i][j] = Mma<DT>::run(a[i], __builtin_bit_cast(frag, ob[kk][j]), acc2[4 * h + i][j]);
                }
            }
        };
        if (first) { __syncthreads(); first = false; }          // BN tables visible
        for (int c = 0; c < NCH; c += 2) {
            chunk(c, std::integral_constant<int, 0>{});
            chunk(c + 1, std::integral_constant<int, 1>{});
        }

        // ---- epilogue 2: pixels 16 j + fr, channels 32 ip + 8 fq .. +8, ip = (NI2W / 2) wc .. -------------------------------------
        F16Guard<DT> guard2;
#pragma unroll
        for (int ipl = 0; ipl < NI2W / 2; ++ipl) {
            const int ch = 32 * ((NI2W / 2) * wc + ipl) + 8 * fq;
            const f32x4 s0 = *reinterpret_cast<const f32x4*>(tsc2 + ch), s1 = *reinterpret_cast<const f32x4*>(tsc2 + ch + 4);
            const f32x4 h0 = *reinterpret_cast<const f32x4*>(tsf2 + ch), h1 = *reinterpret_cast<const f32x4*>(tsf2 + ch + 4);
#pragma unroll
            for (int j = 0; j < PBW; ++j) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc2[2 * ipl][j][e] * s0[e] + h0[e];
                    v[4 + e] = acc2[2 * ipl + 1][j][e] * s1[e] + h1[e];
                }
                apply_act8(v, act2);
                guard2.see(v);
                u32x4 q;
#pragma unroll
                for (int e = 0; e < 4; ++e) q[e] = pack2<DT>(v[2 * e], v[2 * e + 1]);
                const long pix = pix_of(tile, j);
                const uint32_t off = pix < p.M ? (uint32_t)((pix * CM + ch) * 2) : 0x80000000u;
                __builtin_amdgcn_raw_buffer_store_b128(q, y2rsrc, off, 0, 0);
            }
        }
        guard2.commit(p.ovf);
        tile += tstride;
        tile_parity = !tile_parity;
        if (tile >= p.nTiles) break;
        load_x(tile, xf);
    }
    // one weight request is still in flight (made by the last chunk of the last tile): drain before the LDS is released
    wpair_wait_vmcnt<0>();
#endif  // __HIP_DEVICE_COMPILE__
}

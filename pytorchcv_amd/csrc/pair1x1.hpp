// pair1x1.hpp - two back-to-back 1x1 convolutions in one launch: the last convolution of a ResNet bottleneck unit
// (64 -> 256, BN, + identity, ReLU; reference resnet.py:227-228) and the first convolution of the NEXT unit
// (256 -> 64, BN, ReLU; resnet.py:106-109), 16-bit storage.
//
// Run separately, the second launch re-reads the 256-channel tensor the first one has just written (411 MB at
// 56x56, batch 256) - the largest single read of the net. Here a block owns 64 pixels x ALL 256 channels of y1, so
// the freshly rounded y1 values never leave the registers before they are consumed as the second GEMM's operand:
//
//   GEMM1  wave w: y1[64w..64w+63][64 px] = W1[64 rows w][64] . x[64][64 px]       (32 MFMA, W1 slice in 32 VGPRs)
//   epi 1  scale/shift, + residual, ReLU, round to 16 bit -> 16-byte NHWC stores of y1. With the operands swapped
//          (weights = MFMA A) a lane owns 8 consecutive channels of one pixel, which is at the same time exactly the
//          MFMA B fragment of the second GEMM for K-step (2w + ip) and pixel block j: no LDS round trip.
//   GEMM2  wave w: partial y2[64][64 px] over ITS 64 channels of y1 (32 MFMA, W2 column slice in 32 VGPRs)
//   reduce the four K-slices meet in LDS (fp32, fixed order -> deterministic); wave w finishes pixel block j = w
//   epi 2  scale/shift, ReLU, 16-byte stores of y2.
//
// All weights live in registers for the whole persistent block; LDS holds a 3-deep ring of x tiles (8 KB each,
// LDS-DMA with the source-side XOR swizzle) and the 64 KB reduction buffer: one block per CU with the full 512-register
// budget, so latency is hidden by software prefetch: x and the residual two tiles ahead.
//
// IDC variant (first unit of the stage, reference resnet.py:214-228 with resize_identity): the skip tensor is itself a 1x1
// convolution + BN of the unit's input x0 (64 -> 256, no activation). Instead of reading its 256-channel result the block
// recomputes it from the 64-channel x0 tile (a third register-resident weight slice, 16 more MFMA per wave per tile), rounds it
// to the storage type exactly where the separate launch would have stored it, and adds it: the identity convolution's
// launch, its 411 MB write and the 411 MB residual read of this kernel all disappear (at 56x56, batch 256).
#pragma once
#include "pcv_common.hpp"
#include <type_traits>
#include "igemm_conv.hpp"     // Mma<DT>

struct PairParams {
    const void* x;            // [M][64]
    const void* w1;           // packed rows [256][64]  (MFMA row order)
    const void* res;          // [M][256]
    void* y1;                 // [M][256]
    const void* w2;           // packed rows [64][256]
    void* y2;                 // [M][64]
    const float* scale1;
    const float* shift1;
    const float* scale2;
    const float* shift2;
    uint32_t x_bytes, res_bytes, y1_bytes, y2_bytes, w1_bytes, w2_bytes;
    int M, nTiles;
    int act1, post1, act2;
    // IDC: skip = BN_id(W_id . x0) instead of `res`
    const void* x0;           // [M][64]
    const void* wid;          // packed rows [256][64]
    const float* scale_id;
    const float* shift_id;
    uint32_t x0_bytes, wid_bytes;
    // GATE: per-image channel gate on the first convolution (an SE block run inside it), fp32 [N][256]; n = pixel / HW
    const float* gate;
    FastDiv div_hw;
    uint32_t* ovf;            // the context's fp16 overflow counter (pcv_common.hpp, F16Guard)
};

template <int N> __device__ __forceinline__ void pair_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// PB: 16-pixel blocks per tile. PB = 4: 64-pixel tiles, one block per CU (needs > 256 registers); PB = 2: 32-pixel tiles,
// half the accumulators, fits 256 registers and 44 KB of LDS -> two blocks (8 waves) per CU, which hides HBM latency better.
template <int DT, int PB, bool IDC = false, bool GATE = false>
__global__ __launch_bounds__(256, PB == 4 ? 1 : 2) void pair1x1_kernel(const PairParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int K1 = 64, C1 = 256, C2 = 64, P = 16 * PB;
    constexpr int NXQ = PB / 2;                                   // x DMA pieces (8 pixel rows) per wave
    constexpr int NIP = PB / 2;                                   // 32-channel output groups finished per wave
    constexpr int XB = P * K1 * 2;                                // bytes per x tile
    typedef typename Mma<DT>::frag frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [x ring: 3 tiles | (IDC: x0 ring: 3 tiles) | reduction: 16 KB x PB]
    char* const x0ring = smem + 3 * XB;
    char* const red = smem + (IDC ? 6 : 3) * XB;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;

    int tile = blockIdx.x;
    const int tstride = gridDim.x;
    if (tile >= p.nTiles) return;

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res), 0, p.res_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t y1rsrc = __builtin_amdgcn_make_buffer_rsrc(p.y1, 0, p.y1_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t y2rsrc = __builtin_amdgcn_make_buffer_rsrc(p.y2, 0, p.y2_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w1rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w1), 0, p.w1_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w2rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w2), 0, p.w2_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t x0rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(IDC ? p.x0 : p.x), 0, IDC ? p.x0_bytes : p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t widrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(IDC ? p.wid : p.w1), 0, IDC ? p.wid_bytes : p.w1_bytes, 0x00020000);

    // ---- weights -> registers (once) -----------------------------------------------------------------------------
    frag a1[4][2];      // W1 rows 64w + 16i + fr, K-step ks
    frag a2[4][2];      // W2 rows 16i + fr, K-step 2w + ip (this wave's 64 channels of y1)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const u32x4 v1 = __builtin_amdgcn_raw_buffer_load_b128(
                w1rsrc, (uint32_t)(((64 * wave + 16 * i + fr) * K1 + ks * 32 + 8 * fq) * 2), 0, 0);
            const u32x4 v2 = __builtin_amdgcn_raw_buffer_load_b128(
                w2rsrc, (uint32_t)(((16 * i + fr) * C1 + (2 * wave + ks) * 32 + 8 * fq) * 2), 0, 0);
            a1[i][ks] = __builtin_bit_cast(frag, v1);
            a2[i][ks] = __builtin_bit_cast(frag, v2);
        }
    frag aid[IDC ? 4 : 1][2];                                  // IDC: W_id rows 64w + 16i + fr
    if constexpr (IDC) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                aid[i][ks] = __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(
                    widrsrc, (uint32_t)(((64 * wave + 16 * i + fr) * K1 + ks * 32 + 8 * fq) * 2), 0, 0));
    }
    // epilogue constants: channels 64w + 32ip + 8fq + e (first GEMM), 32ip + 8fq + e (second GEMM)
    float sc1[2][8], sf1[2][8], sc2[NIP][8], sf2[NIP][8];
    // IDC: the identity convolution's BN constants live in LDS (2 KB behind the reduction buffer; the registers are full)
    float* const tscid = reinterpret_cast<float*>(red + 16 * 1024 * PB);
    float* const tsfid = tscid + C1;
    float* const tsc1 = tsfid + C1;                             // IDC: the first convolution's constants too (4 KB in all)
    float* const tsf1 = tsc1 + C1;
    if constexpr (IDC) {
        for (int i = tid; i < C1; i += 256) {
            tscid[i] = p.scale_id[i]; tsfid[i] = p.shift_id[i];
            tsc1[i] = p.scale1[i]; tsf1[i] = p.shift1[i];
        }
        __syncthreads();
    }
#pragma unroll
    for (int ip = 0; ip < 2; ++ip)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c1 = 64 * wave + 32 * ip + 8 * fq + e;
            sc1[ip][e] = IDC ? 0.f : p.scale1[c1];
            sf1[ip][e] = IDC ? 0.f : p.shift1[c1];
        }
#pragma unroll
    for (int k = 0; k < NIP; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c2 = 32 * ((wave / PB) * NIP + k) + 8 * fq + e;
            sc2[k][e] = p.scale2[c2];
            sf2[k][e] = p.shift2[c2];
        }
    const ActClamp act1 = make_act(p.act1), post1 = make_act(p.post1), act2 = make_act(p.act2);

    // ---- per-lane constant pieces of the addresses ----------------------------------------------------------------
    // x DMA: wave w moves pieces 2w, 2w+1 (8 pixel rows of 128 B each); lane L -> row 8pc + (L >> 3), LDS slot L & 7,
    // which must receive chunk (slot ^ swz(row)), swz(r) = (r >> 1) & 7 (conflict-free ds_read_b128 over 128-byte rows).
    int xrow[NXQ], xchunk[NXQ];
#pragma unroll
    for (int q = 0; q < NXQ; ++q) {
        const int pc = NXQ * wave + q;
        xrow[q] = 8 * pc + (lane >> 3);
        xchunk[q] = (lane & 7) ^ ((xrow[q] >> 1) & 7);
    }
    auto issue_x = [&](int t, int slot) {
#pragma unroll
        for (int q = 0; q < NXQ; ++q) {
            const long pix = (long)t * P + xrow[q];
            const uint32_t off = (t < p.nTiles && pix < p.M) ? (uint32_t)((pix * K1 + xchunk[q] * 8) * 2) : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, PCV_LDS(smem + slot * XB + (NXQ * wave + q) * 1024), 16, off, 0, 0, 0);
            if constexpr (IDC)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(x0rsrc, PCV_LDS(x0ring + slot * XB + (NXQ * wave + q) * 1024), 16, off, 0, 0, 0);
        }
    };
    // residual / y1 element (ip, j): pixel 16j + fr, channels 64w + 32ip + 8fq .. +8
    auto off_c1 = [&](int t, int ip, int j) -> uint32_t {
        const long pix = (long)t * P + 16 * j + fr;
        return (t < p.nTiles && pix < p.M) ? (uint32_t)((pix * C1 + 64 * wave + 32 * ip + 8 * fq) * 2) : 0x80000000u;
    };
    auto load_res = [&](int t, u32x4 (&r)[2][PB]) {
        if constexpr (IDC) return;                              // the skip tensor is computed, not read
#pragma unroll
        for (int ip = 0; ip < 2; ++ip)
#pragma unroll
            for (int j = 0; j < PB; ++j) r[ip][j] = __builtin_amdgcn_raw_buffer_load_b128(rrsrc, off_c1(t, ip, j), 0, 0);
    };
    // B fragment of GEMM1: pixel row 16j + fr, chunk 4ks + fq
    int xfrag[PB][2];
#pragma unroll
    for (int j = 0; j < PB; ++j)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int row = 16 * j + fr;
            xfrag[j][ks] = row * 128 + (((4 * ks + fq) ^ ((row >> 1) & 7)) << 4);
        }

    // ---- prologue ---------------------------------------------------------------------------------------------------
    // The tile loop is unrolled by 3 = the x ring depth, so that the ring slot and the three residual register sets
    // (current, one ahead, two ahead) are compile-time constants: no register copies, whose operands would force a wait
    // for the prefetched loads at the end of every iteration.
    u32x4 resr[3][2][PB];
    issue_x(tile, 0);
    issue_x(tile + tstride, 1);
    load_res(tile, resr[0]);
    load_res(tile + tstride, resr[1]);
    pair_wait_vmcnt<0>();
    bool first = true;

    F16Guard<DT> guard;
    auto step = [&](auto KC) -> bool {
        constexpr int slot = decltype(KC)::value;
        constexpr int slot2 = (slot + 2) % 3;
        u32x4 (&resc)[2][PB] = resr[slot];
        // x(t) landed: all but the 5 PB youngest VMEM ops of this wave are done - those are the previous iteration's
        // x(t+2) [PB/2], residual(t+2) [2 PB], y1 stores [2 PB], y2 stores [PB/2] (always issued, out of range when invalid).
        // IDC: x(t+2) + x0(t+2) [PB], no residual loads, y1 stores [2 PB], y2 stores [PB/2].
        if (!first) pair_wait_vmcnt<IDC ? (7 * PB) / 2 : 5 * PB>();
        first = false;
        __syncthreads();

        // ---- IDC: the skip tensor of this tile = round(BN_id(W_id . x0)), kept as the packed residual ---------------------
        if constexpr (IDC) {
            f32x4 accd[4][PB];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < PB; ++j) accd[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const char* x0b = x0ring + slot * XB;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                frag b[PB];
#pragma unroll
                for (int j = 0; j < PB; ++j) b[j] = *reinterpret_cast<const frag*>(x0b + xfrag[j][ks]);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < PB; ++j) accd[i][j] = Mma<DT>::run(aid[i][ks], b[j], accd[i][j]);
            }
#pragma unroll
            for (int ip = 0; ip < 2; ++ip) {
                const int ch = 64 * wave + 32 * ip + 8 * fq;
                const f32x4 s0 = *reinterpret_cast<const f32x4*>(tscid + ch), s1 = *reinterpret_cast<const f32x4*>(tscid + ch + 4);
                const f32x4 h0 = *reinterpret_cast<const f32x4*>(tsfid + ch), h1 = *reinterpret_cast<const f32x4*>(tsfid + ch + 4);
#pragma unroll
                for (int j = 0; j < PB; ++j)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const float a0 = accd[2 * ip][j][2 * e] * s0[2 * e] + h0[2 * e];
                        const float a1 = accd[2 * ip][j][2 * e + 1] * s0[2 * e + 1] + h0[2 * e + 1];
                        const float b0 = accd[2 * ip + 1][j][2 * e] * s1[2 * e] + h1[2 * e];
                        const float b1 = accd[2 * ip + 1][j][2 * e + 1] * s1[2 * e + 1] + h1[2 * e + 1];
                        guard.see2(a0, a1);
                        guard.see2(b0, b1);
                        resc[ip][j][e] = pack2<DT>(a0, a1);
                        resc[ip][j][2 + e] = pack2<DT>(b0, b1);
                    }
            }
        }

        // ---- GEMM1 ----------------------------------------------------------------------------------------------------
        f32x4 acc[4][PB];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < PB; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const char* xb = smem + slot * XB;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            frag b[PB];
#pragma unroll
            for (int j = 0; j < PB; ++j) b[j] = *reinterpret_cast<const frag*>(xb + xfrag[j][ks]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < PB; ++j) acc[i][j] = Mma<DT>::run(a1[i][ks], b[j], acc[i][j]);
        }

        // ---- prefetch two tiles ahead: x into the slot read in the previous iteration, residual into the free set ---------
        issue_x(tile + 2 * tstride, slot2);
        load_res(tile + 2 * tstride, resr[slot2]);

        // ---- epilogue 1: BN, + residual, activation, round; the packs are GEMM2's B fragments ---------------------------
        u32x4 o[2][PB];
#pragma unroll
        for (int ip = 0; ip < 2; ++ip)
#pragma unroll
            for (int j = 0; j < PB; ++j) {
                float v[8], r8[8];
                if constexpr (IDC) {
                    const int ch = 64 * wave + 32 * ip + 8 * fq;
                    const f32x4 s0 = *reinterpret_cast<const f32x4*>(tsc1 + ch), s1 = *reinterpret_cast<const f32x4*>(tsc1 + ch + 4);
                    const f32x4 h0 = *reinterpret_cast<const f32x4*>(tsf1 + ch), h1 = *reinterpret_cast<const f32x4*>(tsf1 + ch + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = acc[2 * ip][j][e] * s0[e] + h0[e];
                        v[4 + e] = acc[2 * ip + 1][j][e] * s1[e] + h1[e];
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = acc[2 * ip][j][e] * sc1[ip][e] + sf1[ip][e];
                        v[4 + e] = acc[2 * ip + 1][j][e] * sc1[ip][4 + e] + sf1[ip][4 + e];
                    }
                }
                apply_act8(v, act1);
                if constexpr (GATE) {
#pragma clang fp contract(off)      // as in igemm_conv.hpp: the product is rounded before the skip add
                    const long pix = (long)tile * P + 16 * j + fr;
                    const uint32_t n = fastdiv((uint32_t)(pix < p.M ? pix : p.M - 1), p.div_hw);
                    const float* gp = p.gate + (size_t)n * C1 + 64 * wave + 32 * ip + 8 * fq;
                    const f32x4 g0 = *reinterpret_cast<const f32x4*>(gp), g1 = *reinterpret_cast<const f32x4*>(gp + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] *= g0[e]; v[4 + e] *= g1[e]; }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) unpack2<DT>(resc[ip][j][e], r8[2 * e], r8[2 * e + 1]);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += r8[e];
                apply_act8(v, post1);
                guard.see(v);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[ip][j][e] = pack2<DT>(v[2 * e], v[2 * e + 1]);
            }
#pragma unroll
        for (int ip = 0; ip < 2; ++ip)
#pragma unroll
            for (int j = 0; j < PB; ++j) __builtin_amdgcn_raw_buffer_store_b128(o[ip][j], y1rsrc, off_c1(tile, ip, j), 0, 0);

        // ---- GEMM2, this wave's K slice -----------------------------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < PB; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ip = 0; ip < 2; ++ip)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < PB; ++j)
                    acc[i][j] = Mma<DT>::run(a2[i][ip], __builtin_bit_cast(frag, o[ip][j]), acc[i][j]);

        // ---- the four K slices meet in LDS: [wave][i][j][lane] fp32x4 ------------------------------------------------------
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < PB; ++j)
                *reinterpret_cast<f32x4*>(red + (((wave * 4 + i) * PB + j) * 64 + lane) * 16) = acc[i][j];
        __syncthreads();
        // wave w finishes pixel block jz for NIP of the two 32-channel output groups (PB = 4: one block, both groups;
        // PB = 2: two waves share a pixel block, one group each)
        const int jz = wave % PB, ip0 = (wave / PB) * NIP;
        f32x4 z[2 * NIP];
#pragma unroll
        for (int k = 0; k < 2 * NIP; ++k) {
            const int i = 2 * ip0 + k;
            z[k] = *reinterpret_cast<const f32x4*>(red + (((0 * 4 + i) * PB + jz) * 64 + lane) * 16);
#pragma unroll
            for (int v = 1; v < 4; ++v) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(red + (((v * 4 + i) * PB + jz) * 64 + lane) * 16);
                z[k] += t;
            }
        }
        __syncthreads();                                        // reduction buffer free for the next tile

        // ---- epilogue 2: pixel block jz, channels 32ip + 8fq .. +8 ---------------------------------------------------------------
#pragma unroll
        for (int k = 0; k < NIP; ++k) {
            const int ip = ip0 + k;
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = z[2 * k][e] * sc2[k][e] + sf2[k][e];
                v[4 + e] = z[2 * k + 1][e] * sc2[k][4 + e] + sf2[k][4 + e];
            }
            apply_act8(v, act2);
            guard.see(v);
            u32x4 q;
#pragma unroll
            for (int e = 0; e < 4; ++e) q[e] = pack2<DT>(v[2 * e], v[2 * e + 1]);
            const long pix = (long)tile * P + 16 * jz + fr;
            const uint32_t off = pix < p.M ? (uint32_t)((pix * C2 + 32 * ip + 8 * fq) * 2) : 0x80000000u;
            __builtin_amdgcn_raw_buffer_store_b128(q, y2rsrc, off, 0, 0);
        }
        guard.commit(p.ovf);
        tile += tstride;
        return tile < p.nTiles;
    };
    while (true) {
        if (!step(std::integral_constant<int, 0>{})) break;
        if (!step(std::integral_constant<int, 1>{})) break;
        if (!step(std::integral_constant<int, 2>{})) break;
    }
#endif  // __HIP_DEVICE_COMPILE__
}

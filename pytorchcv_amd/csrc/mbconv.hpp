// mbconv.hpp - a whole inverted-residual unit in one launch: [1x1 expand + BN + act] -> depthwise 3x3 + BN + act ->
// 1x1 project + BN (+ skip add); reference LinearBottleneck.forward (mobilenetv2.py:62-71), MobileNetV3Unit without SE
// (mobilenetv3.py:82-93), DwsConvBlock (conv.py:546-618, no expand). 16-bit storage.
//
// Run as three launches the expanded tensor (6x the unit's input) is written and read twice: 1.2 GB per pass at 112x112,
// batch 512, against 0.2 GB of unit input + output. Here it never leaves the CU. A block owns a TH x 16 tile of output
// pixels and walks the expanded channels in chunks of 32:
//
//   S1  E[chunk][input tile incl. halo]  = act(BN(W_exp[chunk] . x tile))         MFMA, x tile resident in LDS (LDS-DMA),
//                                          pixels outside the image forced to 0  (the depthwise pads the EXPANDED map)
//       without an expand convolution the chunk of x itself is the DMA target.
//   S2  D[chunk][TH x 16 pixels]         = act(BN(depthwise 3x3 of E))            VALU, 9 ds_read_b128 per 8 channels
//   S3  acc[Cout][TH x 16]              += W_proj[:, chunk] . D                   MFMA
//
// then BN (+ residual) and 16-byte NHWC stores. E and D rows are 64 bytes (32 channels) with the 4-slot XOR swizzle
// f = [0,2,3,1][(pixel >> 2) & 3], conflict-free for the MFMA B-fragment reads. Weights of a chunk go global -> registers
// (every lane's fragment is 16 contiguous bytes of the packed blobs). Two barriers per chunk.
#pragma once
#include "pcv_common.hpp"
#include "igemm_conv.hpp"     // Mma<DT>

struct MbParams {
    const void* x;            // [N,H,W,Cin]
    const void* res;          // [N,Ho,Wo,Cout] or null
    void* y;                  // [N,Ho,Wo,Cout]
    const void* w_exp;        // packed rows [round32(Cmid)][Kpad1]        (null pointer semantics: EXPAND == false)
    const void* w_dw;         // [9][Cmid]
    const void* w_proj;       // packed rows [round32(Cout)][Kpad2]
    const float *scale_e, *shift_e, *scale_d, *shift_d, *scale_p, *shift_p;
    uint32_t x_bytes, y_bytes, wexp_bytes, wdw_bytes, wproj_bytes;
    int N, H, W, Cin, Cmid, Cout, Ho, Wo;
    int Kpad1, Kpad2;
    int ka;                   // MFMA K-steps of the expand GEMM = ceil(Cin / 32)
    int nChunks;              // ceil(Cmid / 32)
    int nRowT;                // 16-row tiles of the project GEMM = round32(Cout) / 16  (<= 6)
    int tilesH, tilesW, nTiles;
    int act_e, act_d, act_p, post;
};

__device__ __forceinline__ int mb_swz(int pix) { return (0x78 >> (2 * ((pix >> 2) & 3))) & 3; }   // [0,2,3,1]

// MAXRT: 16-row tiles of the project GEMM held in accumulators (2: Cout <= 32, 6: Cout <= 96)
template <int DT, int S, bool EXPAND, int MAXRT>
__global__ __launch_bounds__(256, 2) void mbconv_kernel(const MbParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int TH = S == 1 ? 8 : 4;                  // output rows per tile (x 16 columns)
    constexpr int IH = (TH - 1) * S + 3, IW = 15 * S + 3;
    constexpr int NIP = IH * IW;                        // input pixels incl. halo: 180 / 297
    constexpr int NPT = (NIP + 15) / 16;                // 16-pixel MFMA blocks over the input tile: 12 / 19
    constexpr int MAXPT = (NPT + 3) / 4;                // per wave
    constexpr int OUTPX = TH * 16;
    constexpr int NJ = TH / 4;                          // output pixel blocks per wave in the project GEMM
    constexpr int NIT = OUTPX * 4 / 256;                // depthwise items (pixel, 8-channel slot) per thread
    typedef typename Mma<DT>::frag frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // [Es: NPT*16 rows x 64 B (two buffers without an expand stage)][Ds: OUTPX x 64 B][valid: NPT*16 bytes, padded]
    // [Xs: ka x NPT*16 x 64 B]
    char* const Es = smem;
    char* const Ds = Es + (EXPAND ? 1 : 2) * NPT * 16 * 64;
    unsigned char* const valid = reinterpret_cast<unsigned char*>(Ds + OUTPX * 64);
    char* const Xs = reinterpret_cast<char*>(valid) + ((NPT * 16 + 15) & ~15);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;

    const int tile = blockIdx.x;
    const int tw = tile % p.tilesW;
    const int t2 = tile / p.tilesW;
    const int th = t2 % p.tilesH;
    const int n = t2 / p.tilesH;
    const int ho0 = th * TH, wo0 = tw * 16;
    const int hi0 = ho0 * S - 1, wi0 = wo0 * S - 1;     // input coordinates of halo pixel (0, 0); padding = 1

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res), 0, p.res ? p.y_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t wersrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w_exp), 0, p.wexp_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wdrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w_dw), 0, p.wdw_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wprsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w_proj), 0, p.wproj_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t sersrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.scale_e), 0, EXPAND ? p.Cmid * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t hersrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.shift_e), 0, EXPAND ? p.Cmid * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t sdrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.scale_d), 0, p.Cmid * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t hdrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.shift_d), 0, p.Cmid * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t sprsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.scale_p), 0, p.Cout * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t hprsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.shift_p), 0, p.Cout * 4, 0x00020000);

    // ---- which input-tile pixels lie inside the image -------------------------------------------------------------------
    for (int ip = tid; ip < NPT * 16; ip += 256) {
        const int r = ip / IW, c = ip - r * IW;
        valid[ip] = (ip < NIP && (unsigned)(hi0 + r) < (unsigned)p.H && (unsigned)(wi0 + c) < (unsigned)p.W) ? 1 : 0;
    }

    // ---- LDS-DMA of 32 channels [c0, c0 + 32) of the input tile into `dst` (rows of 64 B, swizzled): one instruction =
    //      16 pixel rows; lane L -> pixel 16*piece + (L >> 2), physical slot L & 3 which holds chunk (L & 3) ^ f(pixel).
    auto dma_tile = [&](char* dst, int c0) {
        for (int pc = wave; pc < NPT; pc += 4) {
            const int ip = 16 * pc + (lane >> 2);
            const int r = ip / IW, c = ip - r * IW;
            const int ch = c0 + 8 * ((lane & 3) ^ mb_swz(ip));
            const int hi = hi0 + r, wi = wi0 + c;
            const bool ok = ip < NIP && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W && ch < p.Cin;
            const uint32_t off = ok ? (uint32_t)(((((long)n * p.H + hi) * p.W + wi) * p.Cin + ch) * 2) : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, PCV_LDS(dst + pc * 1024), 16, off, 0, 0, 0);
        }
    };
    if constexpr (EXPAND) {
        for (int ks = 0; ks < p.ka; ++ks) dma_tile(Xs + ks * (NPT * 16 * 64), 32 * ks);
    }

    const ActClamp act_e = make_act(p.act_e), act_d = make_act(p.act_d), act_p = make_act(p.act_p), post = make_act(p.post);

    f32x4 acc[MAXRT][NJ];
#pragma unroll
    for (int i = 0; i < MAXRT; ++i)
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) acc[i][jj] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- depthwise work of this thread: 8-channel slot q of output column ocol, NIT vertically adjacent output rows
    //      (stride 1: rows 2g, 2g+1 share a 4-row input window; stride 2: one row). tid = ((g * 16) + ocol) * 4 + q.
    const int q = tid & 3;
    const int ocol = (tid >> 2) & 15;
    const int orow0 = (tid >> 6) * NIT;
    constexpr int WROWS = (NIT - 1) * S + 3;            // input rows of the window
    const int dbase = (orow0 * S) * IW + ocol * S;

    // ---- per-chunk weights: the expand and depthwise weights are re-loaded for the NEXT chunk right after their last use, so
    //      the L2 latency hides under the other stages without a second register set; the BN constants and the project
    //      weights are fetched at the start of the stage before the one that needs them ------------------------------------
    constexpr int KAMAX = 3;
    frag we[2][KAMAX];
    u32x4 wdw[9];
    auto load_w1 = [&](int c) {
        if constexpr (EXPAND) {
            const int ch0 = 32 * c;
            const bool in = c < p.nChunks;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int ks = 0; ks < KAMAX; ++ks)
                    we[i][ks] = __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(
                        wersrc, (in && ks < p.ka) ? (uint32_t)(((ch0 + 16 * i + fr) * p.Kpad1 + 32 * ks + 8 * fq) * 2) : 0x80000000u, 0, 0));
        }
    };
    auto load_w2 = [&](int c) {
        const int ch = 32 * c + 8 * q;
        const bool in = c < p.nChunks && ch < p.Cmid;
#pragma unroll
        for (int t = 0; t < 9; ++t)
            wdw[t] = __builtin_amdgcn_raw_buffer_load_b128(wdrsrc, in ? (uint32_t)((t * p.Cmid + ch) * 2) : 0x80000000u, 0, 0);
    };
    // without an expand convolution the E tile of chunk c is x itself: double-buffered LDS-DMA, one chunk ahead
    constexpr int ESZ = NPT * 16 * 64;
    if constexpr (!EXPAND) dma_tile(Es, 0);
    load_w1(0);
    load_w2(0);
    if constexpr (EXPAND) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // x tile landed (this wave's pieces) ...
        __syncthreads();                                        // ... and everybody else's, and `valid`
    }

    for (int c = 0; c < p.nChunks; ++c) {
        char* const Ec = EXPAND ? Es : Es + (c & 1) * ESZ;
        // ---- S1: E chunk ----------------------------------------------------------------------------------------------------
        if constexpr (EXPAND) {
            f32x4 se[2], he[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t off = (uint32_t)((32 * c + 8 * fq + 4 * h) * 4);
                se[h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(sersrc, off, 0, 0));
                he[h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(hersrc, off, 0, 0));
            }
            f32x4 ea[MAXPT][2];
#pragma unroll
            for (int m = 0; m < MAXPT; ++m) ea[m][0] = ea[m][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KAMAX; ++ks) {
                if (ks < p.ka) {
                    const char* xb = Xs + ks * ESZ;
#pragma unroll
                    for (int m = 0; m < MAXPT; ++m) {
                        const int jt = wave + 4 * m;
                        if (jt < NPT) {
                            const int ip = 16 * jt + fr;
                            const frag b = *reinterpret_cast<const frag*>(xb + ip * 64 + ((fq ^ mb_swz(ip)) << 4));
                            ea[m][0] = Mma<DT>::run(we[0][ks], b, ea[m][0]);
                            ea[m][1] = Mma<DT>::run(we[1][ks], b, ea[m][1]);
                        }
                    }
                }
            }
#pragma unroll
            for (int m = 0; m < MAXPT; ++m) {
                const int jt = wave + 4 * m;
                if (jt < NPT) {
                    const int ip = 16 * jt + fr;
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = ea[m][0][e] * se[0][e] + he[0][e];
                        v[4 + e] = ea[m][1][e] * se[1][e] + he[1][e];
                    }
                    apply_act8(v, act_e);
                    const bool ok = valid[ip] != 0;
                    u32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = ok ? pack2<DT>(v[2 * e], v[2 * e + 1]) : 0u;
                    *reinterpret_cast<u32x4*>(Ec + ip * 64 + ((fq ^ mb_swz(ip)) << 4)) = o;
                }
            }
            load_w1(c + 1);
        } else {
            // DMA(c) was issued one chunk ago, right after barrier 1; younger than it are 19 loads: the previous chunk's BN
            // constants [4] and project weights [6] and this chunk's depthwise weights [9]
            asm volatile("s_waitcnt vmcnt(19)" ::: "memory");
        }
        __syncthreads();
        if constexpr (!EXPAND) {
            if (c + 1 < p.nChunks) dma_tile(Es + ((c + 1) & 1) * ESZ, 32 * (c + 1));     // its buffer was last read in S2(c-1)
        }

        // ---- S2: depthwise 3x3 over the E tile -> D ------------------------------------------------------------------------
        f32x4 sd[2], hd[2];
        frag wp[MAXRT];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t off = (uint32_t)((32 * c + 8 * q + 4 * h) * 4);
            sd[h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(sdrsrc, off, 0, 0));
            hd[h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(hdrsrc, off, 0, 0));
        }
#pragma unroll
        for (int i = 0; i < MAXRT; ++i)
            wp[i] = __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(
                wprsrc, i < p.nRowT ? (uint32_t)(((16 * i + fr) * p.Kpad2 + 32 * c + 8 * fq) * 2) : 0x80000000u, 0, 0));
        {
            f32x2 a[NIT][4];
#pragma unroll
            for (int k = 0; k < NIT; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) a[k][e] = (f32x2){0.f, 0.f};
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                f32x2 wv[3][4];                                  // this column's three taps, unpacked once
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float w0, w1;
                        unpack2<DT>(wdw[dy * 3 + dx][e], w0, w1);
                        wv[dy][e] = (f32x2){w0, w1};
                    }
#pragma unroll
                for (int r = 0; r < WROWS; ++r) {
                    const int ip = dbase + r * IW + dx;
                    const u32x4 ev = *reinterpret_cast<const u32x4*>(Ec + ip * 64 + ((q ^ mb_swz(ip)) << 4));
                    f32x2 xv[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x0, x1;
                        unpack2<DT>(ev[e], x0, x1);
                        xv[e] = (f32x2){x0, x1};
                    }
#pragma unroll
                    for (int k = 0; k < NIT; ++k) {
                        const int dy = r - k * S;                // input row r feeds output row k through tap dy
                        if (dy >= 0 && dy < 3) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) a[k][e] += xv[e] * wv[dy][e];
                        }
                    }
                }
                asm volatile("" ::: "memory");                   // keep only one column of the window in flight (register pressure)
            }
#pragma unroll
            for (int k = 0; k < NIT; ++k) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[2 * e] = a[k][e][0];
                    v[2 * e + 1] = a[k][e][1];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = v[e] * sd[0][e] + hd[0][e];
                    v[4 + e] = v[4 + e] * sd[1][e] + hd[1][e];
                }
                apply_act8(v, act_d);
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = pack2<DT>(v[2 * e], v[2 * e + 1]);
                const int op = (orow0 + k) * 16 + ocol;
                *reinterpret_cast<u32x4*>(Ds + op * 64 + ((q ^ mb_swz(op)) << 4)) = o;
            }
        }
        load_w2(c + 1);
        __syncthreads();

        // ---- S3: project GEMM, K-step = this chunk ------------------------------------------------------------------------------
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
            const int op = 16 * (NJ * wave + jj) + fr;
            const frag b = *reinterpret_cast<const frag*>(Ds + op * 64 + ((fq ^ mb_swz(op)) << 4));
#pragma unroll
            for (int i = 0; i < MAXRT; ++i)
                if (i < p.nRowT) acc[i][jj] = Mma<DT>::run(wp[i], b, acc[i][jj]);
        }
        // (the next chunk's S1 writes E, which nobody reads after the barrier above; its barrier orders S3 against the next S2)
    }

    // ---- epilogue: BN (+ residual), 16-byte NHWC stores ------------------------------------------------------------------------
#pragma unroll
    for (int ipp = 0; ipp < MAXRT / 2; ++ipp) {
        if (2 * ipp < p.nRowT) {
            const int ch = 32 * ipp + 8 * fq;
            f32x4 sp[2], hp[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                sp[h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(sprsrc, (uint32_t)((ch + 4 * h) * 4), 0, 0));
                hp[h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(hprsrc, (uint32_t)((ch + 4 * h) * 4), 0, 0));
            }
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) {
                const int ho = ho0 + NJ * wave + jj, wo = wo0 + fr;
                const bool ok = ch < p.Cout && ho < p.Ho && wo < p.Wo;
                const uint32_t off = ok ? (uint32_t)(((((long)n * p.Ho + ho) * p.Wo + wo) * p.Cout + ch) * 2) : 0x80000000u;
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[2 * ipp][jj][e] * sp[0][e] + hp[0][e];
                    v[4 + e] = acc[2 * ipp + 1][jj][e] * sp[1][e] + hp[1][e];
                }
                apply_act8(v, act_p);
                if (p.res != nullptr) {
                    const u32x4 rv = __builtin_amdgcn_raw_buffer_load_b128(rrsrc, off, 0, 0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float r0, r1;
                        unpack2<DT>(rv[e], r0, r1);
                        v[2 * e] += r0;
                        v[2 * e + 1] += r1;
                    }
                    apply_act8(v, post);
                }
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = pack2<DT>(v[2 * e], v[2 * e + 1]);
                __builtin_amdgcn_raw_buffer_store_b128(o, yrsrc, off, 0, 0);
            }
        }
    }
#endif  // __HIP_DEVICE_COMPILE__
}

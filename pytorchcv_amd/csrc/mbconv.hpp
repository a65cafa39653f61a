// mbconv.hpp - a whole inverted-residual unit in one launch: [1x1 expand + BN + act] -> depthwise 3x3 + BN + act ->
// 1x1 project + BN (+ skip add); reference LinearBottleneck.forward (mobilenetv2.py:62-71), MobileNetV3Unit without SE
// (mobilenetv3.py:82-93), DwsConvBlock (conv.py:546-618, no expand). 16-bit storage.
//
// Run as three launches the expanded tensor (6x the unit's input) is written and read twice: 1.2 GB per pass at 112x112,
// batch 512, against 0.2 GB of unit input + output. Here it never leaves the CU. A persistent block keeps ALL weights and
// BN constants of the unit in LDS (loaded once), owns TH x 16 tiles of output pixels and walks the expanded channels in
// chunks of 32:
//
//   S1  E[chunk][input tile incl. halo]  = act(BN(W_exp[chunk] . x tile))         MFMA, x tile in LDS (LDS-DMA, the next
//                                          tile's x is fetched while this one computes); pixels outside the image are
//                                          forced to 0 (the depthwise pads the EXPANDED map).
//       without an expand convolution the chunk of x itself is the DMA target.
//   S2  D[chunk][TH x 16 pixels]         = act(BN(depthwise 3x3 of E))            MFMA with block-diagonal weights: one
//                                          16x16x32 step = 2 taps x 16 channels (K) -> 16 channels x 16 pixels of one output
//                                          row, 5 steps for the 9 taps; a lane's B chunk is 8 channels of the pixel shifted by
//                                          its tap, read straight from the E tile (gconv3x3.hpp's scheme at group size 1). The
//                                          VALU version of this stage (unpack + v_pk_fma_f32 per tap) was 33-47 % of the kernel.
//   S3  acc[Cout][TH x 16]              += W_proj[:, chunk] . D                   MFMA
//
// then BN (+ residual) and 16-byte NHWC stores. E, D and the weight slabs are rows of 64 bytes (32 channels) with the
// 4-slot XOR swizzle f = [0,2,3,1][(row >> 2) & 3], conflict-free for the MFMA fragment reads. Two barriers per chunk.
// Because the per-chunk operands come from LDS (lgkmcnt), the only vector-memory traffic in flight during a tile is the
// next tile's x, which therefore hides completely.
#pragma once
#include "pcv_common.hpp"
#include "igemm_conv.hpp"     // Mma<DT>

struct MbParams {
    const void* x;            // [N,H,W,Cin]
    const void* res;          // [N,Ho,Wo,Cout] or null
    void* y;                  // [N,Ho,Wo,Cout]
    const void* w_exp;        // packed rows [round32(Cmid)][Kpad1]        (unused when EXPAND == false)
    const void* w_dw;         // [9][Cmid]
    const void* w_dwsp;       // compressed diagonal fragments [chunk][half][filter row][lane] 16 B (pack_dw_sparse_kernel; mbr.hpp)
    const void* w_proj;       // packed rows [round32(Cout)][Kpad2]
    const float *scale_e, *shift_e, *scale_d, *shift_d, *scale_p, *shift_p;
    uint32_t x_bytes, y_bytes, wexp_bytes, wdw_bytes, wproj_bytes;
    int N, H, W, Cin, Cmid, Cout, Ho, Wo;
    int Kpad1, Kpad2;
    int ka;                   // MFMA K-steps of the expand GEMM = ceil(Cin / 32) (<= 3)
    int nChunks;              // ceil(Cmid / 32)
    int nRowT;                // 16-row tiles of the project GEMM = round32(Cout) / 16
    int tilesH, tilesW, nTiles;
    int nbufX;                // x tile buffers (2: the next tile is prefetched)
    int act_e, act_d, act_p, post;
    uint32_t* ovf;            // the context's fp16 overflow counter (pcv_common.hpp, F16Guard)
    uint32_t* dbg;            // diagnostic builds only (-DMBR_CYCLES): in-kernel stamps
};

__device__ __forceinline__ int mb_swz(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }   // [0,2,3,1]

// Host and device agree on the LDS carve-up through this helper (offsets and total in bytes).
struct MbLds {
    int wexp, wproj, wdw, bn, x, e, d, valid, total;
};
static inline __host__ __device__ MbLds mb_lds_layout(int stride, bool expand, int ka, int nChunks, int nRowT, int nbufX) {
    const int npt = stride == 1 ? 12 : 19, outpx = stride == 1 ? 128 : 64;
    MbLds L;
    int o = 0;
    L.wexp = o; o += expand ? nChunks * ka * 32 * 64 : 0;        // [chunk][ks][32 rows][64 B]
    L.wproj = o; o += nChunks * nRowT * 16 * 64;                  // [chunk][nRowT*16 rows][64 B]
    L.wdw = o; o += (9 * nChunks * 32 * 2 + 15) & ~15;            // [9][nChunks*32] 16-bit
    L.bn = o; o += 4 * nChunks * 32 * 4;                          // scale_e, shift_e, scale_d, shift_d: [nChunks*32] fp32 each
    L.x = o; o += expand ? nbufX * ka * npt * 16 * 64 : 0;
    L.e = o; o += (expand ? 1 : 2) * npt * 16 * 64;
    L.d = o; o += outpx * 64;
    L.valid = o; o += (npt * 16 + 15) & ~15;
    L.total = o;
    return L;
}

// MAXRT: 16-row tiles of the project GEMM held in accumulators (2: Cout <= 32, 6: Cout <= 96)
template <int DT, int S, bool EXPAND, int MAXRT>
__global__ __launch_bounds__(256, 2) void mbconv_kernel(const MbParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int TH = S == 1 ? 8 : 4;                  // output rows per tile (x 16 columns)
    constexpr int IH = (TH - 1) * S + 3, IW = 15 * S + 3;
    constexpr int NIP = IH * IW;                        // input pixels incl. halo: 180 / 297
    constexpr int NPT = (NIP + 15) / 16;                // 16-pixel MFMA blocks over the input tile: 12 / 19
    constexpr int MAXPT = (NPT + 3) / 4;                // per wave
    constexpr int NJ = TH / 4;                          // output pixel blocks per wave in the project GEMM
    constexpr int ESZ = NPT * 16 * 64;
    typedef typename Mma<DT>::frag frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const MbLds L = mb_lds_layout(S, EXPAND, p.ka, p.nChunks, p.nRowT, p.nbufX);
    char* const Wes = smem + L.wexp;
    char* const Wps = smem + L.wproj;
    char* const Wds = smem + L.wdw;
    float* const BNs = reinterpret_cast<float*>(smem + L.bn);
    char* const Xs = smem + L.x;
    char* const Es = smem + L.e;
    char* const Ds = smem + L.d;
    unsigned char* const valid = reinterpret_cast<unsigned char*>(smem + L.valid);
    const int CmidP = p.nChunks * 32;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;

    int tile = blockIdx.x;
    if (tile >= p.nTiles) return;

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res), 0, p.res ? p.y_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t wersrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w_exp), 0, p.wexp_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wdrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w_dw), 0, p.wdw_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wprsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w_proj), 0, p.wproj_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t sprsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.scale_p), 0, p.Cout * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t hprsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.shift_p), 0, p.Cout * 4, 0x00020000);

    // ---- the unit's weights -> LDS, once per block -------------------------------------------------------------------------
    // slabs of 16 rows x 64 B per LDS-DMA instruction; lane L -> row 16*piece + (L >> 2), physical slot L & 3, which must
    // receive K chunk (L & 3) ^ f(row)
    if constexpr (EXPAND) {
        const int npieces = p.nChunks * p.ka * 2;                 // [chunk][ks][2 x 16 rows]
        for (int pc = wave; pc < npieces; pc += 4) {
            const int half = pc & 1, ks = (pc >> 1) % p.ka, c = (pc >> 1) / p.ka;
            const int row = 32 * c + 16 * half + (lane >> 2);
            const int kc = (lane & 3) ^ mb_swz(row);
            const uint32_t off = (uint32_t)((row * p.Kpad1 + 32 * ks + 8 * kc) * 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wersrc, PCV_LDS(Wes + pc * 1024), 16, off, 0, 0, 0);
        }
    }
    {
        const int npieces = p.nChunks * p.nRowT;                  // [chunk][nRowT x 16 rows]
        for (int pc = wave; pc < npieces; pc += 4) {
            const int i = pc % p.nRowT, c = pc / p.nRowT;
            const int row = 16 * i + (lane >> 2);
            const int kc = (lane & 3) ^ mb_swz(row);
            const uint32_t off = (uint32_t)((row * p.Kpad2 + 32 * c + 8 * kc) * 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wprsrc, PCV_LDS(Wps + pc * 1024), 16, off, 0, 0, 0);
        }
    }
    for (int i = tid; i < 9 * CmidP / 8; i += 256) {              // depthwise weights, 16 B = 8 channels per item
        const int t = i / (CmidP / 8), ch = (i - t * (CmidP / 8)) * 8;
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wdrsrc, ch < p.Cmid ? (uint32_t)((t * p.Cmid + ch) * 2) : 0x80000000u, 0, 0);
        *reinterpret_cast<u32x4*>(Wds + (t * CmidP + ch) * 2) = v;
    }
    for (int i = tid; i < 4 * CmidP; i += 256) {
        const int which = i / CmidP, ch = i - which * CmidP;
        const float* src = which == 0 ? p.scale_e : which == 1 ? p.shift_e : which == 2 ? p.scale_d : p.shift_d;
        BNs[i] = (ch < p.Cmid && src != nullptr) ? src[ch] : 0.f;
    }

    const ActClamp act_e = make_act(p.act_e), act_d = make_act(p.act_d), act_p = make_act(p.act_p), post = make_act(p.post);

    // ---- LDS-DMA of 32 channels [c0, c0 + 32) of tile t's input window into `dst` (same slab scheme, rows = pixels) ----------
    auto dma_tile = [&](int t, char* dst, int c0) {
        const int tw = t % p.tilesW;
        const int t2 = t / p.tilesW;
        const int th = t2 % p.tilesH;
        const int n = t2 / p.tilesH;
        const int hi0 = th * TH * S - 1, wi0 = tw * 16 * S - 1;
        for (int pc = wave; pc < NPT; pc += 4) {
            const int ip = 16 * pc + (lane >> 2);
            const int r = ip / IW, c = ip - r * IW;
            const int ch = c0 + 8 * ((lane & 3) ^ mb_swz(ip));
            const int hi = hi0 + r, wi = wi0 + c;
            const bool ok = t < p.nTiles && ip < NIP && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W && ch < p.Cin;
            const uint32_t off = ok ? (uint32_t)(((((long)n * p.H + hi) * p.W + wi) * p.Cin + ch) * 2) : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, PCV_LDS(dst + pc * 1024), 16, off, 0, 0, 0);
        }
    };
    auto dma_x = [&](int t, int buf) {
        for (int ks = 0; ks < p.ka; ++ks) dma_tile(t, Xs + (buf * p.ka + ks) * ESZ, 32 * ks);
    };

    // ---- depthwise work of this wave: 16-channel half `dg` of every chunk, NB output rows (16 pixels each) ---------------------
    // A (weights, rows = channels): lane (row fr, k quarter fq) holds k = 8 fq .. 8 fq + 7 = tap (fq >> 1) of the pair, channels
    // 8 (fq & 1) .. + 7 of the half: non-zero only on the diagonal, i.e. element fr & 7 when (fr >> 3) == (fq & 1).
    // B (E tile, columns = pixels of one output row): lane (column fr, fq) reads the same 8 channels of the pixel under its tap.
    constexpr int NB = TH / 2;
    const int dg = wave & 1;
    const int drow0 = (wave >> 1) * NB;
    uint32_t boff[NB][5], doff[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int tap = min(2 * j + (fq >> 1), 8);                   // tap 9 does not exist: its weights are zero
            const int ip = ((drow0 + u) * S + tap / 3) * IW + fr * S + tap % 3;
            boff[u][j] = (uint32_t)(ip * 64 + (((2 * dg + (fq & 1)) ^ mb_swz(ip)) << 4));
        }
        const int op = (drow0 + u) * 16 + fr;                            // accumulator rows 4 fq .. 4 fq + 3 = channels 16 dg + 4 fq ..
        doff[u] = (uint32_t)(op * 64 + (((2 * dg + (fq >> 1)) ^ mb_swz(op)) << 4) + 8 * (fq & 1));
    }
    uint32_t am[4], am4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        am[i] = ((fr >> 3) == (fq & 1) && i == ((fr & 7) >> 1)) ? 0xFFFFFFFFu : 0u;
        am4[i] = (fq >> 1) ? 0u : am[i];
    }
    const int a_sh = (fr & 1) * 16;
    const int a_woff = ((fq >> 1) * CmidP + 16 * dg + fr) * 2;          // + (2 j * CmidP + 32 c) * 2

    if constexpr (EXPAND) dma_x(tile, 0);
    int xbuf = 0;
    const int tstride = gridDim.x;

    while (true) {
        const int tw = tile % p.tilesW;
        const int t2 = tile / p.tilesW;
        const int th = t2 % p.tilesH;
        const int n = t2 / p.tilesH;
        const int ho0 = th * TH, wo0 = tw * 16;
        const int hi0 = ho0 * S - 1, wi0 = wo0 * S - 1;     // input coordinates of halo pixel (0, 0); padding = 1
        const int ntile = tile + tstride;

        // which input-window pixels lie inside the image (safe to overwrite: everybody is past the last S1 of the previous tile)
        for (int ip = tid; ip < NPT * 16; ip += 256) {
            const int r = ip / IW, c = ip - r * IW;
            valid[ip] = (ip < NIP && (unsigned)(hi0 + r) < (unsigned)p.H && (unsigned)(wi0 + c) < (unsigned)p.W) ? 1 : 0;
        }
        if constexpr (!EXPAND) dma_tile(tile, Es, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // weights (first tile), this tile's x / first chunk
        __syncthreads();
        if constexpr (EXPAND) {
            if (p.nbufX == 2) dma_x(ntile, xbuf ^ 1);            // flies during the whole chunk loop (nothing else uses vmcnt)
        }

        f32x4 acc[MAXRT][NJ];
#pragma unroll
        for (int i = 0; i < MAXRT; ++i)
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) acc[i][jj] = (f32x4){0.f, 0.f, 0.f, 0.f};

        for (int c = 0; c < p.nChunks; ++c) {
            F16Guard<DT> g1, g2;                                // fp16 range checks of this chunk's E / D values
            char* const Ec = EXPAND ? Es : Es + (c & 1) * ESZ;
            // ---- S1: E chunk ------------------------------------------------------------------------------------------------
            if constexpr (EXPAND) {
                f32x4 ea[MAXPT][2];
#pragma unroll
                for (int m = 0; m < MAXPT; ++m) ea[m][0] = ea[m][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
                for (int ks = 0; ks < p.ka; ++ks) {
                    frag we[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        we[i] = *reinterpret_cast<const frag*>(Wes + ((c * p.ka + ks) * 32 + 16 * i + fr) * 64 +
                                                               ((fq ^ mb_swz(16 * i + fr)) << 4));
                    const char* xb = Xs + (xbuf * p.ka + ks) * ESZ;
#pragma unroll
                    for (int m = 0; m < MAXPT; ++m) {
                        const int jt = wave + 4 * m;
                        if (jt < NPT) {
                            const int ip = 16 * jt + fr;
                            const frag b = *reinterpret_cast<const frag*>(xb + ip * 64 + ((fq ^ mb_swz(ip)) << 4));
                            ea[m][0] = Mma<DT>::run(we[0], b, ea[m][0]);
                            ea[m][1] = Mma<DT>::run(we[1], b, ea[m][1]);
                        }
                    }
                }
                const f32x4 se0 = *reinterpret_cast<const f32x4*>(BNs + 32 * c + 8 * fq);
                const f32x4 se1 = *reinterpret_cast<const f32x4*>(BNs + 32 * c + 8 * fq + 4);
                const f32x4 he0 = *reinterpret_cast<const f32x4*>(BNs + CmidP + 32 * c + 8 * fq);
                const f32x4 he1 = *reinterpret_cast<const f32x4*>(BNs + CmidP + 32 * c + 8 * fq + 4);
#pragma unroll
                for (int m = 0; m < MAXPT; ++m) {
                    const int jt = wave + 4 * m;
                    if (jt < NPT) {
                        const int ip = 16 * jt + fr;
                        float v[8];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = ea[m][0][e] * se0[e] + he0[e];
                            v[4 + e] = ea[m][1][e] * se1[e] + he1[e];
                        }
                        apply_act8(v, act_e);
                        if (!act_bounded(p.act_e)) g1.see(v);
                        const bool ok = valid[ip] != 0;
                        u32x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = ok ? pack2<DT>(v[2 * e], v[2 * e + 1]) : 0u;
                        *reinterpret_cast<u32x4*>(Ec + ip * 64 + ((fq ^ mb_swz(ip)) << 4)) = o;
                    }
                }
            } else if (c > 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // chunk c of x, issued one chunk ago (nothing younger in flight)
            }
            if (c > 0 || EXPAND) __syncthreads();
            if constexpr (!EXPAND) {
                if (c + 1 < p.nChunks) dma_tile(tile, Es + ((c + 1) & 1) * ESZ, 32 * (c + 1));   // buffer last read in S2(c-1)
            }

            // ---- S2: depthwise 3x3 over the E tile -> D (block-diagonal MFMA) -----------------------------------------------------
            {
                frag af[5];
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    // (the last pair's second tap reads the 16 bits behind the table - still this block's LDS - and is masked off)
                    const uint32_t w16 = *reinterpret_cast<const uint16_t*>(Wds + a_woff + (2 * j * CmidP + 32 * c) * 2);
                    const uint32_t val = w16 << a_sh;
                    u32x4 a4;
#pragma unroll
                    for (int i = 0; i < 4; ++i) a4[i] = val & (j < 4 ? am[i] : am4[i]);
                    af[j] = __builtin_bit_cast(frag, a4);
                }
                const f32x4 sd = *reinterpret_cast<const f32x4*>(BNs + 2 * CmidP + 32 * c + 16 * dg + 4 * fq);
                const f32x4 hd = *reinterpret_cast<const f32x4*>(BNs + 3 * CmidP + 32 * c + 16 * dg + 4 * fq);
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    f32x4 da = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < 5; ++j) {
                        const frag b = *reinterpret_cast<const frag*>(Ec + boff[u][j]);
                        da = Mma<DT>::run(af[j], b, da);
                    }
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = da[e] * sd[e] + hd[e];
                    apply_actn<4>(v, act_d);
                    if (!act_bounded(p.act_d)) g2.see(v);
                    u32x2 o;
                    o[0] = pack2<DT>(v[0], v[1]);
                    o[1] = pack2<DT>(v[2], v[3]);
                    *reinterpret_cast<u32x2*>(Ds + doff[u]) = o;
                }
            }
            __syncthreads();

            g1.commit(p.ovf);
            g2.commit(p.ovf);
            // ---- S3: project GEMM, K-step = this chunk --------------------------------------------------------------------------
            {
                frag wp[MAXRT];
#pragma unroll
                for (int i = 0; i < MAXRT; ++i)
                    if (i < p.nRowT)
                        wp[i] = *reinterpret_cast<const frag*>(Wps + ((c * p.nRowT + i) * 16 + fr) * 64 + ((fq ^ mb_swz(fr)) << 4));
#pragma unroll
                for (int jj = 0; jj < NJ; ++jj) {
                    const int op = 16 * (NJ * wave + jj) + fr;
                    const frag b = *reinterpret_cast<const frag*>(Ds + op * 64 + ((fq ^ mb_swz(op)) << 4));
#pragma unroll
                    for (int i = 0; i < MAXRT; ++i)
                        if (i < p.nRowT) acc[i][jj] = Mma<DT>::run(wp[i], b, acc[i][jj]);
                }
            }
            // (the next chunk's S1 writes E, which nobody reads after the barrier above; its barrier orders S3 against the next S2)
        }

        // ---- epilogue: BN (+ residual), 16-byte NHWC stores --------------------------------------------------------------------
        F16Guard<DT> guard;
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
                    guard.see(v);
                    u32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = pack2<DT>(v[2 * e], v[2 * e + 1]);
                    __builtin_amdgcn_raw_buffer_store_b128(o, yrsrc, off, 0, 0);
                }
            }
        }

        guard.commit(p.ovf);
        if (ntile >= p.nTiles) break;
        tile = ntile;
        if constexpr (EXPAND) {
            if (p.nbufX == 2) {
                xbuf ^= 1;
            } else {
                __syncthreads();                                 // everybody is done reading the single x buffer
                dma_x(tile, 0);
            }
        }
    }
#endif  // __HIP_DEVICE_COMPILE__
}

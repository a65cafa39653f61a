// stem_conv.hpp - first convolution of the nets (Cin <= 4, stride 2): ResNet's 7x7/2 pad 3 (reference resnet.py:250-254) and
// MobileNetV2's 3x3/2 pad 1 init block (mobilenetv2.py:109-113), 16-bit storage, Cout <= 64, fused BN + activation.
//
// The generic implicit GEMM gathers this layer's operand in 16-byte pieces straight from L2 (K = 49 taps x 4 padded
// channels, every input pixel is fetched ~12 times): 1.6 GB of L2->LDS gather traffic for ResNet-50 at batch 256.
// Here the input window of a 16x16 output tile (a 37x40-pixel patch of the zero-padded NHWC4 image, 12 KB) is staged
// in LDS ONCE with coalesced row-contiguous LDS-DMA, and the MFMA B fragments are read from that patch directly:
// for output column `fr` and filter row r, lane group fq needs the 16 bytes (= 2 pixels x 4 channels) at patch pixel
// 2*fr + 2*fq of patch row 2*orow + r - consecutive lanes read consecutive 16-byte slots, so the overlapping windows
// cost nothing. One filter row = one K=32 MFMA step (8 pixels x 4 channels; the leading pad pixel and channel 3 carry
// zero weights). The weights (Cout x kh x 32, <= 28 KB) stay resident in LDS for the whole persistent block.
//
// POOL variant: the MaxPool2d(3, stride 2, pad 1) that follows the stem in the ResNet-style init blocks (resnet.py:255-258)
// in the same launch. A block then owns a 7x7 tile of POOLED pixels = conv rows/cols 14t-1 .. 14t+13 of its 16x16 conv tile
// (1.31x the MFMA work), and the 112x112x64 conv output (411 MB written + read at batch 256) never exists: after BN +
// activation the 3-wide column max runs across lanes with DPP row shifts (a 16-lane DPP row = the 16 conv columns of one
// fragment), the 3-high row max in registers (a wave holds 4 consecutive conv rows) plus one row handed down from the next
// wave through LDS. max() commutes with the monotonic rounding, so the result equals pooling the rounded tensor bit for bit.
//
// NCHW variant (round 2): the kernel reads the caller's fp32 NCHW image itself (the reference's input layout, resnet.py:333) - the
// `pcv_nchw_to_nhwc` launch in front of every net (154 MB read + 103 MB written + 103 MB read again at batch 256) disappears. The
// three fp32 planes of the patch are staged by LDS-DMA (rows of 11 16-byte chunks starting at the 4-pixel boundary at or below
// the patch origin), then 256 threads convert them once per tile into the SAME bf16 / fp16 NHWC4 patch the kernel has always read
// (same rounding as the layout kernel: results are bit-identical), one more barrier per tile; the staging DMA of the next tile
// flies during this tile's MFMAs as before.
#pragma once
#include <type_traits>
#include "pcv_common.hpp"
#include "igemm_conv.hpp"     // Mma<DT>

struct StemParams {
    const void* x;            // NHWC4 [N, H, Wp, 4], Wp even, pad column/channel zero
    const void* w;            // packed [kh][64 rows (MFMA order)][32]  (K element = pixel*4 + channel)
    void* y;                  // NHWC [N, Ho, Wo, Cout]
    const float* scale;
    const float* shift;
    uint32_t x_bytes, w_bytes, y_bytes;
    int N, H, W, Wp, Ho, Wo, Cout;
    int kh;                   // filter rows (kw is folded into the 8-pixel window: kw <= 7)
    int pt;                   // top padding
    int x0off;                // patch column origin relative to 2*wo0 (= -(pl + (pl & 1)))
    int tilesH, tilesW, nTiles;
    int act;
    int Hq, Wq;               // POOL: pooled output size; y is [N, Hq, Wq, Cout]
    int Cin;                  // NCHW variant: planes of x (<= 3); x is fp32 [N, Cin, H, W] with W % 4 == 0, x_bytes its size
    uint32_t* ovf;            // the context's fp16 overflow counter (pcv_common.hpp, F16Guard)
};

static constexpr int kStemStageBytes = 5 * 256 * 16;          // NCHW variant: 3 planes x 37 rows x 11 chunks = 1221 chunks of 16 B

// packed 16-bit values of lane fr + n of the same 16-lane row (0 past the end)
__device__ __forceinline__ uint32_t stem_row_shl_u32(uint32_t v, int n) {
    return (uint32_t)(n == 1 ? __builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xF, 0xF, true) : __builtin_amdgcn_update_dpp(0, (int)v, 0x102, 0xF, 0xF, true));
}
// max of two PACKED pairs of non-negative 16-bit floats (bf16 or fp16): for sign-bit-clear patterns the IEEE order is the unsigned
// integer order (+0 < subnormals < normals < +inf < NaN), so one v_pk_max_u16 is the two float maxima, NaN-propagating like torch's
// max_pool2d; a NaN with the sign bit set is larger still.
__device__ __forceinline__ uint32_t stem_pkmax(uint32_t a, uint32_t b) {
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}

// 256 threads: wave w computes output rows 4w..4w+3 of the 16x16 tile, all 16 CB (padded) channels. CB = 4: up to 64 output
// channels; CB = 2: up to 32 (the 3x3 / 2 stems of MobileNetV2 / V3 / EfficientNet: half the MFMAs and half the epilogue of the
// 64-row form, whose upper 32 rows multiplied zero weights there).
template <int DT, bool POOL = false, bool NCHW = false, int CB = 4>
__global__ __launch_bounds__(256, 2) void stem_conv_kernel(const StemParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int TH = 16, TW = 16;
    constexpr int PCH = 20;                      // 16-byte chunks (pixel pairs) per patch row: 2*15 + 2*3 + 2 -> 20
    constexpr int PCHUNKS = 768;                 // 3 DMA instructions per thread (>= PRMAX * PCH = 740)
    constexpr int WBYTES = 7 * 64 * 64;          // weights: up to 7 filter rows x 64 rows x 64 B
    constexpr int PBYTES = PCHUNKS * 16;
    typedef typename Mma<DT>::frag frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [weights | patch 0 | patch 1 | POOL: row hand-down 6 KB | NCHW: staging]
    constexpr int SFC = 11;                      // NCHW: 16-byte chunks (4 fp32 pixels) per staged plane row: 40 patch pixels + <= 2 of alignment
    char* const stage = smem + WBYTES + 2 * PBYTES + 3 * 2 * 64 * 16;
    constexpr int TSTEP = POOL ? 14 : 16;        // conv rows / columns between tile origins
    constexpr int TORG = POOL ? -1 : 0;          // first conv row / column of tile 0

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int PR = 2 * (TH - 1) + p.kh;

    const int perXcd = (p.nTiles + 7) >> 3;
    const int xcd = blockIdx.x & 7;
    const int tstride = gridDim.x >> 3;
    int tile = xcd * perXcd + (int)(blockIdx.x >> 3);
    const int tend = min(p.nTiles, (xcd + 1) * perXcd);
    if (tile >= tend) return;

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);

    // ---- weights: loaded once. LDS row (r, ch) is 64 B = 4 chunks; chunk slot s holds K-chunk s ^ f(ch) with
    //      f = [0,2,3,1][(ch >> 2) & 3], which makes the 16-lane ds_read_b128 groups conflict-free on 64-byte rows.
    {
        const int nchunks = p.kh * 64 * 4;
        for (int c = tid; c < (nchunks + 255) / 256 * 256; c += 256) {
            const int row = c >> 2, slot = c & 3;
            const int f = (0x78 >> (2 * ((row >> 2) & 3))) & 3;
            const uint32_t off = c < nchunks ? (uint32_t)((row * 4 + (slot ^ f)) * 16) : 0x80000000u;
            char* dst = smem + (c & ~63) * 16;                                   // wave-uniform: 64 consecutive chunks
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, PCV_LDS(dst), 16, off, 0, 0, 0);
        }
    }

    // ---- patch DMA: chunk c = 256*j + tid -> patch row c / PCH, pixel pair c % PCH (tile independent) ------------------
    int prow[3], pcol[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int c = 256 * j + tid;
        prow[j] = c / PCH;
        pcol[j] = c - prow[j] * PCH;
    }
    // NCHW staging: chunk c = 256 j + tid -> (plane, patch row, column chunk) does not depend on the tile: its decomposition (two divisions by
    // constants per chunk) is done once, a tile adds its origin and checks the image border
    int srow[NCHW ? 5 : 1], scol[NCHW ? 5 : 1], soff[NCHW ? 5 : 1];
    uint32_t sok = 0;                                         // bit j: plane and patch row of chunk j exist
    if constexpr (NCHW) {
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int c = 256 * j + tid;
            const int pl = c / (11 * 37), rem = c - pl * (11 * 37);
            const int r = rem / 11, cc = rem - r * 11;
            srow[j] = r;
            scol[j] = 4 * cc;
            soff[j] = ((pl * p.H + r) * p.W + 4 * cc) * 4;
            sok |= (pl < p.Cin && r < PR ? 1u : 0u) << j;
        }
    }
    auto issue_patch = [&](int t, int buf) {
        const int tw = t % p.tilesW;
        const int t2 = t / p.tilesW;
        const int th = t2 % p.tilesH;
        const int n = t2 / p.tilesH;
        const int hi0 = (th * TSTEP + TORG) * 2 - p.pt;
        const int wp0 = (tw * TSTEP + TORG) * 2 + p.x0off;       // even
        if constexpr (NCHW) {
            // chunk c = 256 j + tid -> (plane, patch row, column chunk); source column = 4-aligned origin at or below wp0
            const int wa = wp0 & ~3;
            const int base = ((n * p.Cin * p.H + hi0) * p.W + wa) * 4;       // (plane 0, patch row 0, column chunk 0) of this tile
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const bool ok = ((sok >> j) & 1u) != 0 && (unsigned)(hi0 + srow[j]) < (unsigned)p.H && (unsigned)(wa + scol[j]) < (unsigned)p.W;
                const uint32_t off = ok ? (uint32_t)(base + soff[j]) : 0x80000000u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, PCV_LDS(stage + (256 * j + (tid & ~63)) * 16), 16, off, 0, 0, 0);
            }
            return;
        }
        char* dst0 = smem + WBYTES + buf * PBYTES;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int hi = hi0 + prow[j];
            const int wp = wp0 + 2 * pcol[j];
            const bool ok = prow[j] < PR && (unsigned)hi < (unsigned)p.H && (unsigned)wp < (unsigned)p.Wp;
            const uint32_t off = ok ? (uint32_t)((((n * p.H + hi) * p.Wp + wp)) * 8) : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, PCV_LDS(dst0 + (256 * j + (tid & ~63)) * 16), 16, off, 0, 0, 0);
        }
    };
    F16Guard<DT> guard;                                           // (reset by commit() at the end of every tile)
    // NCHW: staged fp32 planes of tile t -> the 16-bit NHWC4 patch (buffer 0): item c = 256 j + tid = (patch row, pixel pair)
    auto convert_patch = [&](int t) {
        const int tw = t % p.tilesW;
        const int wp0 = (tw * TSTEP + TORG) * 2 + p.x0off;
        const int a = wp0 & 3;                                   // 0 or 2: patch pixel 0 sits `a` floats into the staged row
        const int Wlim = p.W - (wp0 & ~3);                       // staged columns at or beyond this are outside the image
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int col = a + 2 * pcol[j];                     // first of the two pixels, in staged-row floats (even)
            float v[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                const f32x2 f = *reinterpret_cast<const f32x2*>(stage + ((pl * 37 + prow[j]) * SFC * 4 + col) * 4);
                v[0][pl] = f[0];
                v[1][pl] = f[1];
            }
            // (a chunk is wholly inside or wholly outside the image - W % 4 == 0 - and outside chunks were DMA'd as zeros; the
            // limit only matters for the alignment columns of a chunk that straddles nothing: kept for clarity of intent)
            const bool in0 = col < Wlim, in1 = col + 1 < Wlim;
            guard.see(v[0]);
            guard.see(v[1]);
            u32x4 o;
            o[0] = in0 ? pack2<DT>(v[0][0], v[0][1]) : 0u;
            o[1] = in0 ? pack2<DT>(v[0][2], v[0][3]) : 0u;
            o[2] = in1 ? pack2<DT>(v[1][0], v[1][1]) : 0u;
            o[3] = in1 ? pack2<DT>(v[1][2], v[1][3]) : 0u;
            if (256 * j + tid < PCHUNKS && prow[j] < 37)
                *reinterpret_cast<u32x4*>(smem + WBYTES + (256 * j + tid) * 16) = o;
        }
    };

    // ---- fragment addresses ---------------------------------------------------------------------------------------------
    const int wf = (0x78 >> (2 * ((fr >> 2) & 3))) & 3;
    const int wfrag = fr * 64 + ((fq ^ wf) << 4);                // + r*4096 + i*1024
    const int xfrag = (fr + fq) * 16;                            // + (2*orow + r) * PCH*16

    static_assert(CB == 4 || (CB == 2 && !POOL), "the pooled stem (ResNet-style init blocks) has 64 channels");
    float sc[CB / 2][8], sf[CB / 2][8];
#pragma unroll
    for (int ip = 0; ip < CB / 2; ++ip) {
        const int ch0 = 32 * ip + 8 * fq;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bool ok = ch0 + e < p.Cout;
            sc[ip][e] = ok ? (p.scale ? p.scale[ch0 + e] : 1.f) : 0.f;
            sf[ip][e] = ok ? (p.shift ? p.shift[ch0 + e] : 0.f) : 0.f;
        }
    }
    const ActClamp act = make_act(p.act);

    issue_patch(tile, 0);
    int buf = 0;
    while (true) {
        const int ntile = tile + tstride;
        const bool has_next = ntile < tend;
        // The LDS-DMA of this tile's patch (issued one tile ago) is tracked by vmcnt only: __syncthreads() alone is a workgroup
        // fence (lgkmcnt) + barrier, and without this wait the loop had NO vmcnt wait at all - a block's later tiles could read a
        // patch whose last pieces were still in flight (seen once in ~6 full-batch forwards as a few wrong output rows).
        // (Waiting for everything but the previous tile's output stores - a counted vmcnt - measured no faster.)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                         // weights + this tile's patch landed; other buffer free
        if constexpr (NCHW) {
            convert_patch(tile);                                 // staging -> patch 0 (everybody is past the previous tile's reads)
            __syncthreads();                                     // patch complete; staging free for the next tile's planes
        }
        if (has_next) issue_patch(ntile, buf ^ 1);

        f32x4 acc[CB][4];
#pragma unroll
        for (int i = 0; i < CB; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const char* pbase = smem + WBYTES + (NCHW ? 0 : buf) * PBYTES + xfrag;
        auto load_row = [&](int r, frag (&a)[CB], frag (&b)[4]) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < CB; ++i) a[i] = *reinterpret_cast<const frag*>(smem + r * 4096 + i * 1024 + wfrag);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                b[j] = *reinterpret_cast<const frag*>(pbase + (2 * (4 * wave + j) + r) * (PCH * 16));
        };
        // The usual filter heights (7: ResNet-style stems, 3: MobileNet / EfficientNet) as compile-time loops with the next row's fragments
        // requested in front of this row's MFMAs; with a run-time trip count every row waited for its own eight LDS reads.
        auto rows = [&](auto KHc) __attribute__((always_inline)) {
            constexpr int KH = decltype(KHc)::value;
            frag a[2][CB], b[2][4];
            load_row(0, a[0], b[0]);
#pragma unroll
            for (int r = 0; r < KH; ++r) {
                if (r + 1 < KH) load_row(r + 1, a[(r + 1) & 1], b[(r + 1) & 1]);
#pragma unroll
                for (int i = 0; i < CB; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = Mma<DT>::run(a[r & 1][i], b[r & 1][j], acc[i][j]);
            }
        };
        if (p.kh == 7) rows(std::integral_constant<int, 7>{});
        else if (p.kh == 3) rows(std::integral_constant<int, 3>{});
        else
            for (int r = 0; r < p.kh; ++r) {
                frag a[CB], b[4];
                load_row(r, a, b);
#pragma unroll
                for (int i = 0; i < CB; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = Mma<DT>::run(a[i], b[j], acc[i][j]);
            }

        // ---- epilogue: BN + activation, 16-byte NHWC stores (range-checked: tile tails and channel padding drop) ------------
        const int tw = tile % p.tilesW;
        const int t2 = tile / p.tilesW;
        const int th = t2 % p.tilesH;
        const int n = t2 / p.tilesH;
        if constexpr (POOL) {
            // conv position of this lane: row (14 th - 1) + 4 wave + j, column (14 tw - 1) + fr; outside the conv output = -inf
            const int wo = tw * TSTEP + TORG + fr;
            const bool colok = (unsigned)wo < (unsigned)p.Wo;
            char* const hand = smem + WBYTES + 2 * PBYTES;      // [wave 1..3][ip][lane] x 16 bytes
            u32x4 top[2];                                       // this wave's first row after the column max (for the wave above)
            u32x4 mid[2], bot[2];                               // pooled rows 2 wave (rows j = 0..2) and 2 wave + 1 (rows j = 2, 3 + next wave)
            // The fused pool requires ReLU / ReLU6 (pcv_conv2d_maxpool_supported): outputs are >= 0 (or NaN), so round FIRST (max commutes
            // with the monotonic rounding) and pool the PACKED pairs with integer maxima - half the DPP shifts and maxima of an fp32
            // pool, no unpack / repack; positions outside the conv output contribute 0.
            {
#pragma unroll
                for (int ip = 0; ip < 2; ++ip) {
                    uint32_t h[4][4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int ho = th * TSTEP + TORG + 4 * wave + j;
                        const bool ok = colok && (unsigned)ho < (unsigned)p.Ho;
                        float v[8];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = acc[2 * ip][j][e] * sc[ip][e] + sf[ip][e];
                            v[4 + e] = acc[2 * ip + 1][j][e] * sc[ip][4 + e] + sf[ip][4 + e];
                        }
                        clamp8(v, act);
                        if (!act_bounded(p.act)) guard.see(v);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const uint32_t c0 = ok ? pack2<DT>(v[2 * e], v[2 * e + 1]) : 0u;
                            h[j][e] = stem_pkmax(c0, stem_pkmax(stem_row_shl_u32(c0, 1), stem_row_shl_u32(c0, 2)));
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        top[ip][e] = h[0][e];
                        mid[ip][e] = stem_pkmax(h[0][e], stem_pkmax(h[1][e], h[2][e]));
                        bot[ip][e] = stem_pkmax(h[2][e], h[3][e]);
                    }
                    if (wave > 0) *reinterpret_cast<u32x4*>(hand + (((wave - 1) * 2 + ip) * 64 + lane) * 16) = top[ip];
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // the hand-down rows are written; the next patch's DMA stays in flight
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const int qo = tw * 7 + (fr >> 1);
            const bool lane_ok = (fr & 1) == 0 && fr <= 12 && qo < p.Wq;
#pragma unroll
            for (int ip = 0; ip < 2; ++ip) {
                const int ch0 = 32 * ip + 8 * fq;
                if (wave < 3) {
                    const u32x4 nx = *reinterpret_cast<const u32x4*>(hand + ((wave * 2 + ip) * 64 + lane) * 16);
#pragma unroll
                    for (int e = 0; e < 4; ++e) bot[ip][e] = stem_pkmax(bot[ip][e], nx[e]);
                }
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int po = th * 7 + 2 * wave + half;                   // pooled row
                    const bool ok = lane_ok && ch0 < p.Cout && po < p.Hq && (half == 0 || wave < 3);
                    const uint32_t boff = ok ? (uint32_t)(((((n * p.Hq + po) * p.Wq + qo)) * p.Cout + ch0) * 2) : 0x80000000u;
                    __builtin_amdgcn_raw_buffer_store_b128(half == 0 ? mid[ip] : bot[ip], yrsrc, boff, 0, 0);
                }
            }
        } else {
        const int wo = tw * TW + fr;
#pragma unroll
        for (int ip = 0; ip < CB / 2; ++ip) {
            const int ch0 = 32 * ip + 8 * fq;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ho = th * TH + 4 * wave + j;
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[2 * ip][j][e] * sc[ip][e] + sf[ip][e];
                    v[4 + e] = acc[2 * ip + 1][j][e] * sc[ip][4 + e] + sf[ip][4 + e];
                }
                apply_act8(v, act);
                if (!act_bounded(p.act)) guard.see(v);
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = pack2<DT>(v[2 * e], v[2 * e + 1]);
                const bool ok = ch0 < p.Cout && ho < p.Ho && wo < p.Wo;
                const uint32_t boff = ok ? (uint32_t)(((((n * p.Ho + ho) * p.Wo + wo)) * p.Cout + ch0) * 2) : 0x80000000u;
                __builtin_amdgcn_raw_buffer_store_b128(o, yrsrc, boff, 0, 0);
            }
        }
        }
        guard.commit(p.ovf);
        if (!has_next) break;
        tile = ntile;
        buf ^= 1;
    }
#endif  // __HIP_DEVICE_COMPILE__
}

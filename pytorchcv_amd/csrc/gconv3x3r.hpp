// gconv3x3r.hpp - grouped 3x3 / pad 1 convolution on ROW tiles: stride 2 with 4..32 channels per group and stride 1 with 32
// channels per group (the first unit of ResNeXt / SE-ResNeXt stages 2-4 and all of stage 4; reference resnext.py:62-80 via
// conv3x3_block(stride=strides, groups=cardinality), common/conv.py:340-386), 16-bit storage. Companion of gconv3x3.hpp (stride 1,
// 4..16 channels per group, flat 128-pixel tiles), which these layers could not use: a flat output range of a stride-2 layer needs
// an input window of 4x its pixels plus whole halo rows, and 32 channels per group do not fit the (2 taps x 16 channels) K-step.
//   * a block owns 64 channels (whole groups) x R whole OUTPUT ROWS (R Wo <= 64 pixels = one 16-pixel MFMA block per wave). With an
//     even H the input rows of global output row g (= n Ho + ho) are the flat rows S g - 1 .. S g + 1 whatever the image, so a tile's
//     window is ONE contiguous pixel range ((S R + 3 - S) W pixels + 2) staged once by LDS-DMA, double buffered; image borders are
//     resolved by zeroing the B fragment of a lane whose tap leaves the image;
//   * stride 2: the window is stored split by COLUMN PARITY (even pixels in LDS rows [0, XH), odd pixels in [XH, 2 XH)): the 16
//     lanes of a fragment read (output pixels wo .. wo + 15 -> input columns 2 wo + q - 1) then sit in consecutive 128-byte rows
//     of one region and the row-XOR swizzle stays conflict-free exactly as at stride 1 (unsplit, rows two apart collide 4-way);
//   * KT = 5: K-step = (2 taps) x (16 channels of the slab), weights block-diagonal over the slab's groups (gconv3x3.hpp's blob);
//     KT = 9: K-step = (1 tap) x (the 32 input channels of the group): no zero padding at all, two slabs share each B fragment;
//   * all weights of the 64-channel block stay in registers while consecutive tiles keep the channel block.
// HBM-bound by construction: every input pixel's 128-byte segment is read once per tile (+ 1 halo row in 2 R + 1, L2 / MALL hits).
#pragma once
#include "gconv3x3.hpp"

struct GConvRParams {
    const void* x;
    const void* w;          // KT = 5: gconv3x3.hpp's blob; KT = 9: [C / 16 slabs][9 taps][16 rows][32 input channels of the group]
    void* y;
    const float* scale;
    const float* shift;
    uint32_t x_bytes, w_bytes, y_bytes;
    int Min, Mout;          // input / output pixels of the batch
    int W, Wo, Ho, C;
    int R, RWo;             // output rows per tile, R * Wo (<= 64)
    int XH;                 // stride 2: LDS rows of one column-parity region (a multiple of 8)
    int xl;                 // LDS-DMA pieces per thread: a tile buffer has 32 xl rows of 128 bytes
    int win;                // input pixels of a tile's window
    FastDiv div_wo, div_ho;
    int nRowTiles, nTiles;
    int act;
    uint32_t* ovf;          // the context's fp16 overflow counter (pcv_common.hpp, F16Guard)
};

template <int DT, int S, int KT>
__global__ __launch_bounds__(256, 2) void gconv3x3r_kernel(const GConvRParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(DT != PCV_F32 && (S == 1 || S == 2) && (KT == 5 || KT == 9), "16-bit storage; stride 1 or 2; tap pairs or single taps");
    typedef typename Mma<DT>::frag frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [X: 2 x (32 xl) rows of 128 B]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane >> 3;
    const int cs_lane = (lane & 7) ^ lrow;
    const int fr = lane & 15, fq = lane >> 4;
    const int thalf = fq >> 1, chalf = fq & 1;

    const int perXcd = (p.nTiles + 7) >> 3;
    const int xcd = blockIdx.x & 7;
    const int tstride = gridDim.x >> 3;
    int tile = xcd * perXcd + (int)(blockIdx.x >> 3);
    const int tend = min(p.nTiles, (xcd + 1) * perXcd);
    if (tile >= tend) return;

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    const int xbuf = p.xl * (32 * 128);

    // tile t = (channel block t / nRowTiles, output rows R (t % nRowTiles) .. + R); window pixel c' = c - cbase lives in LDS row
    // c' (stride 1) or (c' >> 1) + (c' & 1) XH (stride 2: cbase is even, so the parity of c' is the parity of the column)
    auto issue_x = [&](int t, int xb, bool live) {
        char* xdst = smem + xb * xbuf;
        const int cb = t / p.nRowTiles;
        const int g0 = (t - cb * p.nRowTiles) * p.R;
        const int cbase = (S * g0 - 1) * p.W - S;
        for (int j = 0; j < p.xl; ++j) {
            const int rho = 8 * (j * 4 + wave) + lrow;
            const int cw = S == 2 ? (rho < p.XH ? 2 * rho : 2 * (rho - p.XH) + 1) : rho;
            const int c = cbase + cw;
            const uint32_t off = (live && cw < p.win && c >= 0 && c < p.Min) ? (uint32_t)((c * p.C + cb * 64 + cs_lane * 8) * 2) : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, PCV_LDS(xdst + (8 * (j * 4 + wave)) * 128), 16, off, 0, 0, 0);
        }
    };

    // this lane's output pixel of the tile (the same for every tile: tiles are whole rows) and the LDS row of its tap (r, q):
    // rbase + r rstep + qoff(q)
    const int ml = wave * 16 + fr;
    const bool lvalid = ml < p.RWo;
    const int mlc = lvalid ? ml : p.RWo - 1;
    const int ho_l = (int)fastdiv((uint32_t)mlc, p.div_wo);
    const int wo = mlc - ho_l * p.Wo;
    const int rbase = S == 2 ? 1 + ho_l * p.W + wo : ho_l * p.W + wo;
    const int rstep = S == 2 ? p.W >> 1 : p.W;
    const int qo0 = S == 2 ? p.XH - 1 : 0, qo1 = S == 2 ? 0 : 1, qo2 = S == 2 ? p.XH : 2;
    const bool at_lo = wo == 0;
    const bool at_hi = S == 1 && wo == p.W - 1;        // stride 2, even W: column 2 wo + 1 is always inside
    const ActClamp act = make_act(p.act);

    frag a[4][KT];                                     // [slab of the channel block][K-step]
    f32x4 sc[4], sf[4];
    int cur_cb = -1;

    issue_x(tile, 0, true);
    int xb = 0;
    while (true) {
        const int ntile = tile + tstride;
        const bool has_next = ntile < tend;
        const int cb = tile / p.nRowTiles;
        const int g0 = (tile - cb * p.nRowTiles) * p.R;
        bool reloaded = false;
        if (cb != cur_cb) {                            // uniform: new channel block -> its weights and BN constants
            cur_cb = cb;
            reloaded = true;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int ks = 0; ks < KT; ++ks)
                    a[i][ks] = __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(
                        wrsrc, (uint32_t)(((((cb * 4 + i) * KT + ks) * 16 + fr) * 32 + 8 * fq) * 2), 0, 0));
                const int ch = cb * 64 + 16 * i + 4 * fq;
                sc[i] = p.scale ? *reinterpret_cast<const f32x4*>(p.scale + ch) : (f32x4){1.f, 1.f, 1.f, 1.f};
                sf[i] = p.shift ? *reinterpret_cast<const f32x4*>(p.shift + ch) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        const uint32_t grow = (uint32_t)(g0 + ho_l);   // global output row -> row inside its image
        const int gh = (int)(grow - fastdiv(grow, p.div_ho) * (uint32_t)p.Ho);
        const bool at_top = gh == 0;
        const bool at_bot = S == 1 && gh == p.Ho - 1;  // stride 2, even H: row 2 ho + 1 is always inside
        // X(tile) landed: the only younger VMEM ops of this wave are the previous tile's 4 stores (a weight reload waits for all)
        if (reloaded) gconv_wait_vmcnt<0>();
        else gconv_wait_vmcnt<4>();
        __builtin_amdgcn_s_barrier();                  // ... for every wave; the other buffer is no longer read
        asm volatile("" ::: "memory");
        issue_x(ntile, xb ^ 1, has_next);

        f32x4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const char* xtile = smem + xb * xbuf;
        auto tap_row = [&](int r, int q) { return rbase + r * rstep + (q == 0 ? qo0 : (q == 1 ? qo1 : qo2)); };
        auto tap_kill = [&](int r, int q) { return (r == 0 && at_top) || (r == 2 && at_bot) || (q == 0 && at_lo) || (q == 2 && at_hi); };
        if constexpr (KT == 5) {
#pragma unroll
            for (int ks = 0; ks < 5; ++ks) {
                // this lane's tap of the pair (2 ks, 2 ks + 1); tap 9 does not exist (zero weights): read tap 8 instead
                const int t0 = 2 * ks, t1 = 2 * ks + 1 < 9 ? 2 * ks + 1 : 8;
                const int row = thalf ? tap_row(t1 / 3, t1 % 3) : tap_row(t0 / 3, t0 % 3);
                const bool kill = thalf ? tap_kill(t1 / 3, t1 % 3) : tap_kill(t0 / 3, t0 % 3);
                const char* xrow = xtile + row * 128;
                const int rsw = row & 7;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    frag b = *reinterpret_cast<const frag*>(xrow + (((2 * i + chalf) ^ rsw) << 4));
                    if (kill) b = (frag){};
                    acc[i] = Mma<DT>::run(a[i][ks], b, acc[i]);
                }
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < 9; ++ks) {
                const int row = tap_row(ks / 3, ks % 3);
                const bool kill = tap_kill(ks / 3, ks % 3);
                const char* xrow = xtile + row * 128;
                const int rsw = row & 7;
#pragma unroll
                for (int g = 0; g < 2; ++g) {          // group g of the channel block: slabs 2 g, 2 g + 1 read the same 32 channels
                    frag b = *reinterpret_cast<const frag*>(xrow + (((4 * g + fq) ^ rsw) << 4));
                    if (kill) b = (frag){};
                    acc[2 * g] = Mma<DT>::run(a[2 * g][ks], b, acc[2 * g]);
                    acc[2 * g + 1] = Mma<DT>::run(a[2 * g + 1][ks], b, acc[2 * g + 1]);
                }
            }
        }

        // ---- epilogue: MFMA rows 4 fq + e of slab i = channels 64 cb + 16 i + 4 fq + e, output pixel g0 Wo + ml: 8-byte stores ----
        const int m = g0 * p.Wo + ml;
        const bool ok = lvalid && m < p.Mout;
        F16Guard<DT> guard;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ch = cb * 64 + 16 * i + 4 * fq;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[i][e] * sc[i][e] + sf[i][e];
            apply_actn(v, act);
            guard.see(v);
            u32x2 o = {pack2<DT>(v[0], v[1]), pack2<DT>(v[2], v[3])};
            __builtin_amdgcn_raw_buffer_store_b64(o, yrsrc, ok ? (uint32_t)((m * p.C + ch) * 2) : 0x80000000u, 0, 0);
        }
        guard.commit(p.ovf, ok);                       // (a lane without an output pixel multiplies whatever its LDS rows hold)
        if (!has_next) break;
        tile = ntile;
        xb ^= 1;
    }
    gconv_wait_vmcnt<0>();                             // the look-ahead DMA of the last tile (issued out of range)
#endif  // __HIP_DEVICE_COMPILE__
}

// ---- weight packing, 32 channels per group: w fp32 [C][32][3][3] -> [C / 16][9][16][32]: row = output channel of the slab, K = the
//      32 input channels of its group at tap ks --------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(256) void pack_gconv32_kernel(const float* __restrict__ w, void* __restrict__ out, int C) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)(C / 16) * 9 * 16 * 32;
    if (i >= total) return;
    const int k = (int)(i & 31);
    const int row = (int)((i >> 5) & 15);
    const int ks = (int)((i >> 9) % 9);
    const int slab = (int)((i >> 9) / 9);
    const int o = slab * 16 + row;
    store_elem<DT>(out, (size_t)i, w[((size_t)o * 32 + k) * 9 + ks]);
}

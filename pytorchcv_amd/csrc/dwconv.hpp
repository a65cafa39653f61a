// dwconv.hpp - depthwise KSxKS convolution, NHWC, fused scale/shift/activation/residual epilogue.
//
// Replaces: dwconv_block -> ConvBlock(groups=out_channels) (reference pytorchcv/models/common/conv.py:437-473),
//           i.e. Conv2d(groups=C) + BatchNorm2d(eval) + ReLU6 of LinearBottleneck.conv2 (mobilenetv2.py:53-57).
//
// HBM-bound (3.4 FLOP/B): the job is to read every input byte once and write every output byte once with 16-byte
// accesses and enough of them in flight. One thread owns 8 consecutive channels (one 16-byte NHWC chunk at 16 bit) of
// one output column and walks DOWN the image with a KS x KS fp32 window in registers, so per output row it loads only
// the STRIDE new input rows (KS chunks each).
//   * Loads are `buffer_load_dwordx4` with range checking: a padded (out-of-image) tap is an offset beyond
//     num_records and returns zeros - no branches, so all loads of a step issue back to back.
//   * Software prefetch: the rows needed by output row h+1 are requested before the FMAs of row h.
//   * The window rotates by register renaming (the row loop is unrolled over the rotation period), no moves.
//   * Consecutive lanes = consecutive channel chunks, then consecutive output columns: every wave-level access is one
//     contiguous NHWC span; the +-1 column re-reads hit the same lines in the vector L1.
#pragma once
#include <type_traits>
#include <utility>
#include "pcv_common.hpp"

struct DwParams {
    const void* x;
    const void* w;        // packed [KS*KS][C] in DT
    const void* res;
    void* y;
    const float* scale;
    const float* shift;
    uint32_t x_bytes;     // buffer num_records of x
    int N, H, W, C, Ho, Wo;
    int pt, pl;
    int C8;               // C / channels-per-thread
    int TH;               // output rows per thread
    int nseg;             // ceil(Ho / TH)
    int act, post_act;
    long total;           // N * nseg * Wo * C8
    int flags;            // tuning: bit 0 = non-temporal output stores, bit 1 = blocks in dispatch order (no XCD remap)
    uint32_t* ovf;        // the context's fp16 overflow counter (pcv_common.hpp, F16Guard)
};

template <int DT> __device__ __forceinline__ void load8(const void* base, size_t eidx, float (&v)[8]) {
    if constexpr (DT == PCV_F32) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + eidx);
        const f32x4 b = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + eidx + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
    } else {
        const u32x4 r = *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint16_t*>(base) + eidx);
#pragma unroll
        for (int e = 0; e < 4; ++e) unpack2<DT>(r[e], v[2 * e], v[2 * e + 1]);
    }
}
template <int DT> __device__ __forceinline__ void store8(void* base, size_t eidx, const float (&v)[8], bool nt = false) {
    if constexpr (DT == PCV_F32) {
        float* p = reinterpret_cast<float*>(base) + eidx;
        *reinterpret_cast<f32x4*>(p) = (f32x4){v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(p + 4) = (f32x4){v[4], v[5], v[6], v[7]};
    } else {
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pack2<DT>(v[2 * e], v[2 * e + 1]);
        u32x4* q = reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(base) + eidx);
        if (nt) __builtin_nontemporal_store(o, q);      // (launch-uniform) streaming store: the line is not kept for a reader that never comes
        else *q = o;
    }
}

// raw CPT-channel chunk as it comes from memory (CPT = 8: 16 B at 16 bit / 32 B fp32; CPT = 4: 8 B / 16 B). The dwords stay in
// the vector registers the load wrote (a scalar array here costs the stride-1 3x3 kernel 14 %: measured).
template <int ND> struct RawVec;
template <> struct RawVec<8> { u32x4 q[2]; __device__ __forceinline__ uint32_t get(int i) const { return q[i >> 2][i & 3]; } };
template <> struct RawVec<4> { u32x4 q[1]; __device__ __forceinline__ uint32_t get(int i) const { return q[0][i]; } };
template <> struct RawVec<2> { u32x2 q[1]; __device__ __forceinline__ uint32_t get(int i) const { return q[0][i]; } };
template <int DT, int CPT> struct RawChunk : RawVec<(DT == PCV_F32 ? CPT : CPT / 2)> {};

template <int DT, int CPT>
__device__ __forceinline__ void raw_load(const __amdgpu_buffer_rsrc_t& rsrc, uint32_t off, RawChunk<DT, CPT>& r) {
    constexpr int ND = DT == PCV_F32 ? CPT : CPT / 2;     // dwords
    if constexpr (ND == 8) {
        r.q[0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
        r.q[1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 16, 0);
    } else if constexpr (ND == 4) {
        r.q[0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
    } else {
        r.q[0] = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off, 0, 0);
    }
}

// CPT channels as CPT/2 packed pairs: the FMAs below lower to v_pk_fma_f32 (two lanes of fp32 per instruction)
template <int DT, int CPT> __device__ __forceinline__ void raw_to_f32x2(const RawChunk<DT, CPT>& r, f32x2 (&v)[CPT / 2]) {
#pragma unroll
    for (int e = 0; e < CPT / 2; ++e) {
        if constexpr (DT == PCV_F32) {
            v[e] = (f32x2){__uint_as_float(r.get(2 * e)), __uint_as_float(r.get(2 * e + 1))};
        } else {
            float lo, hi;
            unpack2<DT>(r.get(e), lo, hi);
            v[e] = (f32x2){lo, hi};
        }
    }
}

// load / store CPT consecutive elements as fp32 (plain pointers: weights, scale/shift, residual, output)
template <int DT, int CPT> __device__ __forceinline__ void loadn(const void* base, size_t eidx, float (&v)[CPT]) {
    if constexpr (CPT == 8) {
        load8<DT>(base, eidx, v);
    } else if constexpr (DT == PCV_F32) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + eidx);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = a[e];
    } else {
        const u32x2 r = *reinterpret_cast<const u32x2*>(reinterpret_cast<const uint16_t*>(base) + eidx);
        unpack2<DT>(r[0], v[0], v[1]);
        unpack2<DT>(r[1], v[2], v[3]);
    }
}
template <int DT, int CPT> __device__ __forceinline__ void storen(void* base, size_t eidx, const float (&v)[CPT], bool nt = false) {
    if constexpr (CPT == 8) {
        store8<DT>(base, eidx, v, nt);
    } else if constexpr (DT == PCV_F32) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + eidx) = (f32x4){v[0], v[1], v[2], v[3]};
    } else {
        u32x2 o = {pack2<DT>(v[0], v[1]), pack2<DT>(v[2], v[3])};
        *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(base) + eidx) = o;
    }
}

// acc += a * b on two fp32 lanes per instruction. hipcc scalarises most `__builtin_elementwise_fma` on float2 here, so the
// instruction is named directly (pure register-to-register, no memory, no hazards with the surrounding VALU code).
__device__ __forceinline__ void pk_fma_acc(f32x2& acc, const f32x2& a, const f32x2& b) {
    asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ f32x2 pk_mul(const f32x2& a, const f32x2& b) {
    f32x2 r;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <int... I, typename F> __device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

// FAST: both activations are none/relu/relu6 (a clamp); the general activation codes live in the FAST=false build so
// that their transcendental code does not bloat the hot kernel.
// CPT: channels per thread (8, or 4 for the 5x5 window whose 25 taps x 8 channels would not fit the register file).
template <int DT, int KS, int S, bool FAST, int CPT = 8>
__global__ __launch_bounds__(256, 2) void dwconv_kernel(const DwParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int ES = Elem<DT>::BYTES;
    constexpr int KEEP = KS > S ? KS - S : 0;       // rows shared by consecutive output rows
    constexpr int NEW = KS - KEEP;                  // rows fetched per output row
    // Neighbouring blocks read overlapping input columns (+-1 column per output column: up to three blocks per line). Blocks b and
    // b + 8 share an XCD's L2, not b and b + 1: in dispatch order every shared line was fetched into two or three L2s (PMC: 1.33x the
    // algorithmic bytes at the fabric, which this HBM-bound kernel cannot afford). xcd_remap gives each XCD a contiguous run of blocks.
    const long idx = (long)((p.flags & 2) ? blockIdx.x : xcd_remap(blockIdx.x, gridDim.x)) * 256 + threadIdx.x;      // (flags bit 1: dispatch order, A/B)
    if (idx >= p.total) return;
    const int c8 = (int)(idx % p.C8);
    long t = idx / p.C8;
    const int wo = (int)(t % p.Wo);
    t /= p.Wo;
    const int seg = (int)(t % p.nseg);
    const int n = (int)(t / p.nseg);
    const int c0 = c8 * CPT;
    const int ho_begin = seg * p.TH;
    const int ho_end = min(p.Ho, ho_begin + p.TH);

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);

    constexpr int NV = CPT / 2;          // packed fp32 pairs per chunk
    f32x2 wgt[KS * KS][NV];
#pragma unroll
    for (int k = 0; k < KS * KS; ++k) {
        float w8[CPT];
        loadn<DT, CPT>(p.w, (size_t)k * p.C + c0, w8);
#pragma unroll
        for (int e = 0; e < NV; ++e) wgt[k][e] = (f32x2){w8[2 * e], w8[2 * e + 1]};
    }
    f32x2 sc[NV], sf[NV];
    {
        float a8[CPT], b8[CPT];
        loadn<PCV_F32, CPT>(p.scale, c0, a8);
        loadn<PCV_F32, CPT>(p.shift, c0, b8);
#pragma unroll
        for (int e = 0; e < NV; ++e) { sc[e] = (f32x2){a8[2 * e], a8[2 * e + 1]}; sf[e] = (f32x2){b8[2 * e], b8[2 * e + 1]}; }
    }
    const ActClamp act = make_act(p.act), pact = make_act(p.post_act);

    // per-column byte offsets (relative to the row start) or "invalid"
    const int wi0 = wo * S - p.pl;
    uint32_t coloff[KS];
#pragma unroll
    for (int q = 0; q < KS; ++q)
        coloff[q] = (unsigned)(wi0 + q) < (unsigned)p.W ? (uint32_t)(((wi0 + q) * p.C + c0) * ES) : 0x80000000u;
    const uint32_t rowbytes = (uint32_t)(p.W * p.C * ES);
    const uint32_t imgoff = (uint32_t)n * (uint32_t)p.H * rowbytes;

    auto fetch_row = [&](int hi, RawChunk<DT, CPT> (&row)[KS]) {
        const bool rowok = (unsigned)hi < (unsigned)p.H;
        const uint32_t rbase = imgoff + (uint32_t)hi * rowbytes;
#pragma unroll
        for (int q = 0; q < KS; ++q) {
            const uint32_t off = (rowok && coloff[q] != 0x80000000u) ? rbase + coloff[q] : 0x80000000u;
            raw_load<DT, CPT>(xrsrc, off, row[q]);
        }
    };

    f32x2 win[KS][KS][NV];               // slot-indexed rows; row r of the current window lives in slot (r + S*phase) % KS
    constexpr int PD = (KS == 3 && S == 1) ? 3 : 1;   // prefetch depth in output rows; divides the unroll period (static ring slots)
    RawChunk<DT, CPT> raw[PD][NEW][KS];       // ring: raw[t % PD] holds the NEW rows of output row t (relative to ho_begin)

    // prologue: the KEEP rows shared with the first output row go straight into the window; the NEW rows of the first
    // PD output rows are requested up front
    const int hi_first = ho_begin * S - p.pt;
    {
        RawChunk<DT, CPT> tmp[KEEP > 0 ? KEEP : 1][KS];
#pragma unroll
        for (int r = 0; r < KEEP; ++r) fetch_row(hi_first + r, tmp[r]);
#pragma unroll
        for (int d = 0; d < PD; ++d) {
            // rows beyond this thread's strip are not requested (offset forced invalid through hi = -1)
            const bool need = ho_begin + d < ho_end;
#pragma unroll
            for (int r = 0; r < NEW; ++r) fetch_row(need ? hi_first + d * S + KEEP + r : -1, raw[d][r]);
        }
#pragma unroll
        for (int r = 0; r < KEEP; ++r)
#pragma unroll
            for (int q = 0; q < KS; ++q) raw_to_f32x2<DT, CPT>(tmp[r][q], win[r][q]);
    }

    int ho = ho_begin;
    int hi = hi_first;
    F16Guard<DT> guard;
    const bool bounded = act_bounded(p.post_act) || (p.post_act == PCV_ACT_NONE && p.res == nullptr && act_bounded(p.act));
    while (ho < ho_end) {
        static_for<KS>([&](auto PHC) {
            constexpr int PH = decltype(PHC)::value;
            if (ho < ho_end) {
                // rows KEEP..KS-1 of this window arrive from the ring slot of this output row
#pragma unroll
                for (int r = 0; r < NEW; ++r)
#pragma unroll
                    for (int q = 0; q < KS; ++q) raw_to_f32x2<DT, CPT>(raw[PH % PD][r][q], win[(KEEP + r + S * PH) % KS][q]);
                // refill the slot with the rows of output row ho + PD before doing this row's arithmetic
                {
                    const bool need = ho + PD < ho_end;
#pragma unroll
                    for (int r = 0; r < NEW; ++r) fetch_row(need ? hi + PD * S + KEEP + r : -1, raw[PH % PD][r]);
                }
                f32x2 acc[NV];
#pragma unroll
                for (int e = 0; e < NV; ++e) acc[e] = pk_mul(win[(S * PH) % KS][0][e], wgt[0][e]);
#pragma unroll
                for (int r = 0; r < KS; ++r)
#pragma unroll
                    for (int q = 0; q < KS; ++q) {
                        if (r == 0 && q == 0) continue;
#pragma unroll
                        for (int e = 0; e < NV; ++e)
                            pk_fma_acc(acc[e], win[(r + S * PH) % KS][q][e], wgt[r * KS + q][e]);
                    }

                const size_t eoff = (((size_t)n * p.Ho + ho) * p.Wo + wo) * p.C + c0;
                float v[CPT];
#pragma unroll
                for (int e = 0; e < NV; ++e) {
                    f32x2 t2 = sf[e];
                    pk_fma_acc(t2, acc[e], sc[e]);
                    v[2 * e] = t2[0];
                    v[2 * e + 1] = t2[1];
                }
                if constexpr (FAST) clampn(v, act); else apply_actn(v, act);
                if (p.res != nullptr) {
                    float r8[CPT];
                    loadn<DT, CPT>(p.res, eoff, r8);
#pragma unroll
                    for (int e = 0; e < CPT; ++e) v[e] += r8[e];
                }
                if (p.post_act != PCV_ACT_NONE) {
                    if constexpr (FAST) clampn(v, pact); else apply_actn(v, pact);
                }
                if (!bounded) guard.see(v);
                storen<DT, CPT>(p.y, eoff, v, (p.flags & 1) != 0);
                ++ho;
                hi += S;
            }
        });
    }
    guard.commit(p.ovf);
#endif  // __HIP_DEVICE_COMPILE__
}

// ---- depthwise 5x5 (dwconv5x5_block, conv.py:511-543: MobileNetV3, EfficientNet), row-streaming formulation -----------------------
// The register-window scheme above needs KS*KS input chunks AND KS*KS weights live per thread: 200 registers at 5x5 even with 4
// channels per thread, which spilled. Here a thread keeps only ONE input row of its 5 columns and accumulates it into the (up to)
// five output rows it touches: input row t of the strip feeds output row o = (t - dy) / S through filter row dy. Live state: 25
// weights + 5 (stride 1) or 3 (stride 2) accumulators + one row = ~170 registers, nothing spills, the same 5 loads per output row.
// One thread = 4 channels (8 bytes at 16 bit) of one output column over its strip of rows, like dwconv_kernel<..., 4>.
template <int DT, int S, bool FAST>
__global__ __launch_bounds__(256, 2) void dwconv5_kernel(const DwParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int KS = 5, CPT = 4, NV = 2;
    constexpr int ES = Elem<DT>::BYTES;
    constexpr int NSLOT = S == 1 ? 5 : 3;          // output rows in flight
    constexpr int PERIOD = S == 1 ? 5 : 6;         // input rows after which the (phase -> slot) pattern repeats
    const long idx = (long)xcd_remap(blockIdx.x, gridDim.x) * 256 + threadIdx.x;      // (see dwconv_kernel)
    if (idx >= p.total) return;
    const int c4 = (int)(idx % p.C8);
    long t0 = idx / p.C8;
    const int wo = (int)(t0 % p.Wo);
    t0 /= p.Wo;
    const int seg = (int)(t0 % p.nseg);
    const int n = (int)(t0 / p.nseg);
    const int c0 = c4 * CPT;
    const int ho_begin = seg * p.TH;
    const int nrows = min(p.Ho, ho_begin + p.TH) - ho_begin;

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);

    f32x2 wgt[KS * KS][NV];
#pragma unroll
    for (int k = 0; k < KS * KS; ++k) {
        float w4[CPT];
        loadn<DT, CPT>(p.w, (size_t)k * p.C + c0, w4);
#pragma unroll
        for (int e = 0; e < NV; ++e) wgt[k][e] = (f32x2){w4[2 * e], w4[2 * e + 1]};
    }
    f32x2 sc[NV], sf[NV];
    {
        float a4[CPT], b4[CPT];
        loadn<PCV_F32, CPT>(p.scale, c0, a4);
        loadn<PCV_F32, CPT>(p.shift, c0, b4);
#pragma unroll
        for (int e = 0; e < NV; ++e) { sc[e] = (f32x2){a4[2 * e], a4[2 * e + 1]}; sf[e] = (f32x2){b4[2 * e], b4[2 * e + 1]}; }
    }
    const ActClamp act = make_act(p.act), pact = make_act(p.post_act);
    F16Guard<DT> guard;
    const bool bounded = act_bounded(p.post_act) || (p.post_act == PCV_ACT_NONE && p.res == nullptr && act_bounded(p.act));

    const int wi0 = wo * S - p.pl;
    uint32_t coloff[KS];
#pragma unroll
    for (int q = 0; q < KS; ++q)
        coloff[q] = (unsigned)(wi0 + q) < (unsigned)p.W ? (uint32_t)(((wi0 + q) * p.C + c0) * ES) : 0x80000000u;
    const uint32_t rowbytes = (uint32_t)(p.W * p.C * ES);
    const uint32_t imgoff = (uint32_t)n * (uint32_t)p.H * rowbytes;
    auto fetch_row = [&](int hi, bool need, RawChunk<DT, CPT> (&row)[KS]) {
        const bool rowok = need && (unsigned)hi < (unsigned)p.H;
        const uint32_t rbase = imgoff + (uint32_t)hi * rowbytes;
#pragma unroll
        for (int q = 0; q < KS; ++q)
            raw_load<DT, CPT>(xrsrc, (rowok && coloff[q] != 0x80000000u) ? rbase + coloff[q] : 0x80000000u, row[q]);
    };

    f32x2 acc[NSLOT][NV];
#pragma unroll
    for (int s = 0; s < NSLOT; ++s)
#pragma unroll
        for (int e = 0; e < NV; ++e) acc[s][e] = (f32x2){0.f, 0.f};

    const int hi_first = ho_begin * S - p.pt;
    const int nsteps = (nrows - 1) * S + KS;       // input rows this strip touches
    // input rows live in a ring with one slot per phase of the unroll period: the row of step t sits in ring[t % PERIOD] and the
    // row PD steps ahead is requested into its own (static) slot - no register copies, nothing waits on the newest loads
    constexpr int PD = 2;
    RawChunk<DT, CPT> ring[PERIOD][KS];
#pragma unroll
    for (int d = 0; d < PD; ++d) fetch_row(hi_first + d, d < nsteps, ring[d]);
    int t = 0;
    while (t < nsteps) {
        static_for<PERIOD>([&](auto PHC) {
            constexpr int PH = decltype(PHC)::value;
            if (t < nsteps) {
                fetch_row(hi_first + t + PD, t + PD < nsteps, ring[(PH + PD) % PERIOD]);
                f32x2 xv[KS][NV];
#pragma unroll
                for (int q = 0; q < KS; ++q) raw_to_f32x2<DT, CPT>(ring[PH][q], xv[q]);
#pragma unroll
                for (int dy = 0; dy < KS; ++dy) {
                    const int d = PH - dy;                                 // S * (output row), modulo the unroll period
                    if (S == 1 || (d & 1) == 0) {                          // stride 2: this input row only meets every other filter row
                        const int slot = (((S == 1 ? d : d / 2) % NSLOT) + NSLOT) % NSLOT;
                        if (t - dy >= 0) {
#pragma unroll
                            for (int q = 0; q < KS; ++q)
#pragma unroll
                                for (int e = 0; e < NV; ++e) pk_fma_acc(acc[slot][e], xv[q][e], wgt[dy * KS + q][e]);
                        }
                    }
                }
                // the output row whose last filter row (dy = 4) was this input row is complete
                {
                    const int d = PH - (KS - 1);
                    if (S == 1 || (d & 1) == 0) {
                        const int slot = (((S == 1 ? d : d / 2) % NSLOT) + NSLOT) % NSLOT;
                        const int o2 = t - (KS - 1);
                        if (o2 >= 0) {
                            const int o = o2 / S;
                            if (o < nrows) {
                                const size_t eoff = (((size_t)n * p.Ho + ho_begin + o) * p.Wo + wo) * p.C + c0;
                                float v[CPT];
#pragma unroll
                                for (int e = 0; e < NV; ++e) {
                                    f32x2 t2 = sf[e];
                                    pk_fma_acc(t2, acc[slot][e], sc[e]);
                                    v[2 * e] = t2[0];
                                    v[2 * e + 1] = t2[1];
                                }
                                if constexpr (FAST) clampn(v, act); else apply_actn(v, act);
                                if (p.res != nullptr) {
                                    float r4[CPT];
                                    loadn<DT, CPT>(p.res, eoff, r4);
#pragma unroll
                                    for (int e = 0; e < CPT; ++e) v[e] += r4[e];
                                }
                                if (p.post_act != PCV_ACT_NONE) {
                                    if constexpr (FAST) clampn(v, pact); else apply_actn(v, pact);
                                }
                                if (!bounded) guard.see(v);
                                storen<DT, CPT>(p.y, eoff, v, (p.flags & 1) != 0);
                            }
#pragma unroll
                            for (int e = 0; e < NV; ++e) acc[slot][e] = (f32x2){0.f, 0.f};
                        }
                    }
                }
                ++t;
            }
        });
    }
    guard.commit(p.ovf);
#endif  // __HIP_DEVICE_COMPILE__
}


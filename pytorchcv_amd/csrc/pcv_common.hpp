// pcv_common.hpp - shared device helpers for the gfx950 kernels (element types, activations, fast division).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/pcv_amd.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define PCV_LDS(p) ((__attribute__((address_space(3))) void*)(p))

// ---- element traits: storage type tag -> bytes, conversions -------------------------------------------
template <int DT> struct Elem;
template <> struct Elem<PCV_F32> { static constexpr int BYTES = 4; };
template <> struct Elem<PCV_BF16> { static constexpr int BYTES = 2; };
template <> struct Elem<PCV_F16> { static constexpr int BYTES = 2; };

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t b16) { return __uint_as_float(b16 << 16); }

// fp32 -> bf16 (round-to-nearest-even; a plain cast lowers to v_cvt_pk_bf16_f32 and keeps NaN a NaN)
__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
    typedef __attribute__((ext_vector_type(2))) float f2;
    f2 v = {lo, hi};
    bf2 r = __builtin_convertvector(v, bf2);
    return *reinterpret_cast<uint32_t*>(&r);
}
__device__ __forceinline__ uint32_t pack2_f16(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) _Float16 h2;
    typedef __attribute__((ext_vector_type(2))) float f2;
    f2 v = {lo, hi};
    h2 r = __builtin_convertvector(v, h2);
    return *reinterpret_cast<uint32_t*>(&r);
}
// fp16 only: round the pair FIRST, then clamp the packed pair to [0, upper] (upper = two fp16 values, e.g. 0x46004600 = 6.0) with
// v_pk_maximum3_f16 / v_pk_minimum3_f16 - two instructions for two values where the fp32 clamp takes four. Same bits as clamping
// in fp32 and rounding afterwards: rounding is monotonic and 0 / 6 are fp16 values (a value beyond fp16's range rounds to +-inf and
// is clamped to upper / 0 like its fp32 original); both forms propagate NaN (IEEE-754-2019 maximum / minimum).
__device__ __forceinline__ uint32_t pack2_clamp_f16(float lo, float hi, uint32_t upper) {
    typedef __attribute__((ext_vector_type(2))) _Float16 h2;
    typedef __attribute__((ext_vector_type(2))) float f2;
    f2 v = {lo, hi};
    h2 r = __builtin_convertvector(v, h2);
    const h2 zero = {(_Float16)0.f, (_Float16)0.f};
    r = __builtin_elementwise_maximum(r, zero);
    r = __builtin_elementwise_minimum(r, __builtin_bit_cast(h2, upper));
    return __builtin_bit_cast(uint32_t, r);
}
template <int DT> __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    if constexpr (DT == PCV_BF16) return pack2_bf16(lo, hi);
    else return pack2_f16(lo, hi);
}
// unpack the two 16-bit elements of a dword
template <int DT> __device__ __forceinline__ void unpack2(uint32_t w, float& lo, float& hi) {
    if constexpr (DT == PCV_BF16) {
        lo = __uint_as_float(w << 16);
        hi = __uint_as_float(w & 0xFFFF0000u);
    } else {
        typedef __attribute__((ext_vector_type(2))) _Float16 h2;
        h2 v = *reinterpret_cast<h2*>(&w);
        lo = (float)v[0];
        hi = (float)v[1];
    }
}
template <int DT> __device__ __forceinline__ float load_elem(const void* base, size_t idx) {
    if constexpr (DT == PCV_F32) return reinterpret_cast<const float*>(base)[idx];
    else if constexpr (DT == PCV_BF16) return bf16_bits_to_f32(reinterpret_cast<const uint16_t*>(base)[idx]);
    else return (float)reinterpret_cast<const _Float16*>(base)[idx];
}
template <int DT> __device__ __forceinline__ void store_elem(void* base, size_t idx, float v) {
    if constexpr (DT == PCV_F32) reinterpret_cast<float*>(base)[idx] = v;
    else if constexpr (DT == PCV_BF16) reinterpret_cast<uint16_t*>(base)[idx] = (uint16_t)(pack2_bf16(v, 0.f) & 0xFFFFu);
    else reinterpret_cast<_Float16*>(base)[idx] = (_Float16)v;
}

// ---- fp16 range guard ------------------------------------------------------------------------------------
// fp16 storage rounds |v| >= 65520 to infinity, and a ReLU6 / sigmoid / h-sigmoid behind it clamps that infinity back into
// range: an overflow inside a net would come out as plausible finite logits. So every kernel that rounds fp32 results to fp16
// checks the magnitudes it rounds (one v_max3_f32 per two values; NaNs are ignored - they propagate by themselves) and bumps the context's overflow counter when one crossed fp16's range. pcv_fp16_guard_begin / _end
// (include/pcv_amd.h) turn a counter change during a forward into NaN logits. Compiled out for bf16 / fp32 storage.
//   A guard object is STAGE-LOCAL: declared where a kernel starts rounding a batch of results (an epilogue, one chunk of a fused
//   unit) and `commit()`ted at the end of that stage - one atomic from one lane if anything crossed the range. Two forms, measured
//   per kernel (MobileNetV3-large, fp16 against bf16, rocprofv3):
//     SCALAR = false: a per-lane running maximum in ONE vector register, one wave ballot at commit (4 VALU per eight values). The
//       default: generic / stem / depthwise / elementwise kernels (igemm h-swish layers +5 % against +9 % with the scalar form).
//     SCALAR = true: the verdict of every group is a wave ballot OR-ed into a scalar register pair - no vector state at all, one
//       compare more per group. For the register-bound fused-unit kernels (mbw.hpp: 233-256 registers): there the extra vector
//       register of the other form spilled inside the chunk loop (the wide dynamic-activation instance +45 % instead of +5 %).
//   (Declared at kernel scope instead of per stage, either form lived across the K loops and spilled 140-320 bytes.)
constexpr float kF16Overflow = 65520.f;
template <int OT, bool SCALAR = false> struct F16Guard {
    float m;
    unsigned long long hit;
    __device__ __forceinline__ F16Guard() : m(0.f), hit(0ull) {}
    __device__ __forceinline__ void note(float t) {
#ifndef PCV_NO_F16_GUARD      // timing experiments only (make EXTRA=-DPCV_NO_F16_GUARD): what the range check costs
        if constexpr (SCALAR) hit |= __builtin_amdgcn_ballot_w64(t >= kF16Overflow);
        else m = fmaxf(m, t);
#endif
    }
    __device__ __forceinline__ void see2(float a, float b) {
        if constexpr (OT == PCV_F16) note(fmaxf(fabsf(a), fabsf(b)));
    }
    template <int N> __device__ __forceinline__ void see(const float (&v)[N]) {
        static_assert(N % 2 == 0, "pairs");
        if constexpr (OT == PCV_F16) {
            float t = fmaxf(fabsf(v[0]), fabsf(v[1]));
#pragma unroll
            for (int e = 2; e < N; e += 2) t = fmaxf(fmaxf(fabsf(v[e]), fabsf(v[e + 1])), t);
            note(t);
        }
    }
    // as see(), for a lane that may hold garbage which is never stored (`live` false): branch-free - the maximum of a dead lane counts as 0.
    // (A divergent `if (live) see(v)` around the scalar form's ballot made mbr_kernel<fp16, launch-time activation> compute wrong rows:
    // built without the guard, or with this form, it is bit-exact; tests/test_gpu_blocks.py, the "keep" shapes.)
    template <int N> __device__ __forceinline__ void see_if(const float (&v)[N], bool live) {
        static_assert(N % 2 == 0, "pairs");
        if constexpr (OT == PCV_F16) {
            float t = fmaxf(fabsf(v[0]), fabsf(v[1]));
#pragma unroll
            for (int e = 2; e < N; e += 2) t = fmaxf(fmaxf(fabsf(v[e]), fabsf(v[e + 1])), t);
            note(live ? t : 0.f);
        }
    }
    // `live`: false for a lane whose values are never stored and may be garbage (per-lane form only)
    __device__ __forceinline__ void commit(uint32_t* counter, bool live = true) {
        if constexpr (OT == PCV_F16) {
            const unsigned long long h = SCALAR ? hit : __builtin_amdgcn_ballot_w64(live && m >= kF16Overflow);
            if (h != 0ull && counter != nullptr) {
                const unsigned active = (unsigned)__builtin_ctzll(__builtin_amdgcn_ballot_w64(true));      // one lane of those here
                if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == active) atomicAdd(counter, 1u);
            }
            m = 0.f;
            hit = 0ull;
        }
    }
};

// ---- activations (reference: pytorchcv/models/common/activ.py) ------------------------------------------
__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case PCV_ACT_RELU: return __builtin_elementwise_maximum(v, 0.f);           // activ.py:64 (NaN stays NaN, as in torch)
        case PCV_ACT_RELU6: return __builtin_elementwise_minimum(__builtin_elementwise_maximum(v, 0.f), 6.f);   // activ.py:81
        case PCV_ACT_SIGMOID: return __builtin_amdgcn_rcpf(1.f + __expf(-v));        // activ.py:132 (v_rcp_f32: 1 ulp)
        case PCV_ACT_SWISH: return v * __builtin_amdgcn_rcpf(1.f + __expf(-v));      // activ.py:20-21
        case PCV_ACT_HSIGMOID: return v != v ? v : fminf(fmaxf(v + 3.f, 0.f), 6.f) * (1.f / 6.f);      // activ.py:29-30
        case PCV_ACT_HSWISH: return v * fminf(fmaxf(v + 3.f, 0.f), 6.f) * (1.f / 6.f);    // activ.py:46-47
        default: return v;
    }
}
// The common activations (none / relu / relu6) with launch-uniform codes: no activation is no instruction, ReLU is one
// v_maximum3_f32, ReLU6 adds one v_minimum3_f32 - gfx950's IEEE-754-2019 maximum / minimum, which PROPAGATE NaN like torch's
// relu / hardtanh (v_max_f32 / v_med3_f32 return the non-NaN operand, which would turn a NaN accumulator - corrupt weights,
// overflowed activations - into 0 and hide it from every isfinite check downstream). One uniform branch per group, none per element.
struct ActClamp {
    float lo, hi;
    bool slow;
    int code;
};
// true when a value that went through `act` (and nothing unbounded after it) lies in [0, 6]: the fp16 range check is skipped
__device__ __forceinline__ bool act_bounded(int act) { return act == PCV_ACT_RELU6 || act == PCV_ACT_SIGMOID || act == PCV_ACT_HSIGMOID; }
__device__ __forceinline__ ActClamp make_act(int act) {
    ActClamp a;
    a.code = act;
    a.slow = act > PCV_ACT_RELU6;
    a.lo = (act == PCV_ACT_RELU || act == PCV_ACT_RELU6) ? 0.f : -INFINITY;
    a.hi = act == PCV_ACT_RELU6 ? 6.f : INFINITY;
    return a;
}
template <int N> __device__ __forceinline__ void clampn(float (&v)[N], const ActClamp& a) {
    if (a.code == PCV_ACT_NONE) return;
    if (a.code == PCV_ACT_RELU) {
#pragma unroll
        for (int e = 0; e < N; ++e) v[e] = __builtin_elementwise_maximum(v[e], 0.f);
    } else {
#pragma unroll
        for (int e = 0; e < N; ++e) v[e] = __builtin_elementwise_minimum(__builtin_elementwise_maximum(v[e], 0.f), 6.f);
    }
}
// the non-clamp activations: ONE uniform switch around a straight unrolled run per case (a switch per element costs eight
// scalar compare/branch sequences in every epilogue group)
template <int N> __device__ __forceinline__ void apply_slow_actn(float (&v)[N], int code) {
    switch (code) {
        case PCV_ACT_SIGMOID:
#pragma unroll
            for (int e = 0; e < N; ++e) v[e] = __builtin_amdgcn_rcpf(1.f + __expf(-v[e]));
            break;
        case PCV_ACT_SWISH:
#pragma unroll
            for (int e = 0; e < N; ++e) v[e] = v[e] * __builtin_amdgcn_rcpf(1.f + __expf(-v[e]));
            break;
        case PCV_ACT_HSIGMOID:
#pragma unroll
            for (int e = 0; e < N; ++e) v[e] = v[e] != v[e] ? v[e] : fminf(fmaxf(v[e] + 3.f, 0.f), 6.f) * (1.f / 6.f);
            break;
        case PCV_ACT_HSWISH:
#pragma unroll
            for (int e = 0; e < N; ++e) v[e] = v[e] * fminf(fmaxf(v[e] + 3.f, 0.f), 6.f) * (1.f / 6.f);
            break;
        default:
#pragma unroll
            for (int e = 0; e < N; ++e) v[e] = apply_act(v[e], code);
    }
}
template <int N> __device__ __forceinline__ void apply_actn(float (&v)[N], const ActClamp& a) {
    if (a.slow) apply_slow_actn<N>(v, a.code);
    else clampn<N>(v, a);
}
__device__ __forceinline__ void clamp8(float (&v)[8], const ActClamp& a) { clampn<8>(v, a); }
__device__ __forceinline__ void apply_act8(float (&v)[8], const ActClamp& a) { apply_actn<8>(v, a); }

// ---- exact unsigned division by a launch-time constant, n < 2^31 ----------------------------------------
struct FastDiv {
    uint32_t d, magic, shift;
};
static inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    f.d = d;
    if (d <= 1) { f.magic = 0; f.shift = 0; return f; }
    uint32_t sh = 0;
    while ((1ull << (sh + 1)) < d) ++sh;          // 2^sh < d <= 2^(sh+1)
    f.shift = sh;
    f.magic = (uint32_t)(((1ull << (32 + sh)) + d - 1) / d);
    return f;
}
__device__ __forceinline__ uint32_t fastdiv(uint32_t n, const FastDiv& f) {
    return f.d <= 1 ? n : (__umulhi(n, f.magic) >> f.shift);
}

// XCD-aware logical tile id: blocks b and b+8 share an XCD (observed round-robin dispatch), so give each XCD a
// contiguous run of logical tiles. Bijective for any tile count (cdna_hip_programming.md section 5, T1).
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t nwg) {
    const uint32_t q = nwg >> 3, r = nwg & 7u, xcd = b & 7u, idx = b >> 3;
    const uint32_t start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + idx;
}

// Dev micro-benchmark (round 4), ping-pong VARIANTS: where the LDS-DMA pieces are issued. The K loop of a SELF-LOADING 8-wave dense 3x3 tile - every wave issues its share of the
// LDS-DMA pieces, reads its fragments and runs its MFMAs; two waves per SIMD, 256 registers, one block per CU - at the real
// footprint of a 256 ch x 224 px (or 128 x 448) block tile with filter-row reuse: 3-deep weight ring, 2 activation slots,
// swizzled 128-byte rows, padded-tap selects. No tile schedule, no epilogue: the hardware's rate for this loop shape.
//   MODE = where a wave issues its pieces (ping-pong of two wave groups, {reads | MFMAs}, four barriers per K-step):
//   0: behind the fragment reads of its read interval (the d3w_kernel order)   1: spread between the MFMAs of its MFMA interval
//   2: in front of its MFMAs   3: behind its MFMAs   4: in front of the fragment reads   5: no pieces at all (ring filled once)
// The activation rows move on every 36 K-steps (a new tile), so that they come from the Infinity Cache / HBM like in a layer.
// hipcc -O3 --offload-arch=gfx950 pingpong_dma_loop.cpp -o pingpong_dma_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <type_traits>
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) char lds_char;
typedef const __attribute__((address_space(3))) s16x8* lds_fptr;

template <int WC, int WP, int CBW, int PBW> struct Cfg {
    static constexpr int BM = 16 * CBW * WC, BP = 16 * PBW * WP;
    static constexpr int NPA = BM / 8, WLW = NPA / 8;
    static constexpr int NPB = (BP + 2 + 7) / 8, XLW = (NPB + 7) / 8;
    static constexpr int ASZ = BM * 128, BSZ = NPB * 1024;
    static constexpr int ZOFF = (3 * ASZ + 2 * BSZ + 2047) / 2048 * 2048, DUMP = ZOFF + 2048, LDS = DUMP + 1024;
    static_assert(LDS <= 160 * 1024, "LDS");
};

__device__ __forceinline__ void sync() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

template <int MODE, int GRP, int WC, int WP, int CBW, int PBW>
__device__ __forceinline__ void body(char* smem, const int wave, const char* __restrict__ Wt, const char* __restrict__ X, float* __restrict__ out, int nk, int Kpad,
                                            int Cin, int Wimg, int M, uint32_t hm0, uint32_t hm2) {
    typedef Cfg<WC, WP, CBW, PBW> G;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wc = wave / WP, wp = wave % WP, fr = lane & 15, fq = lane >> 4;
    const int lrow = lane >> 3, cs = (lane & 7) ^ lrow;
    const uint32_t lds0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(size_t)(lds_char*)smem);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Wt), 0, (uint32_t)(G::BM * Kpad * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(X), 0, (uint32_t)((size_t)M * Cin * 2), 0x00020000);
    for (int i = tid; i < G::LDS / 16; i += 512) reinterpret_cast<f32x4*>(smem)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    // DMA tables
    uint32_t woff[G::WLW], xoff[G::XLW];
#pragma unroll
    for (int i = 0; i < G::WLW; ++i) woff[i] = (uint32_t)(((8 * (8 * i + wave) + lrow) * Kpad + cs * 8) * 2);
    const int P0 = (int)blockIdx.x * G::BP;
#pragma unroll
    for (int j = 0; j < G::XLW; ++j) {
        const int u = 8 * (8 * j + wave) + lrow;
        int m = P0 + u - 1;
        m = m < 0 ? 0 : (m >= M ? M - 1 : m);
        xoff[j] = (uint32_t)((m * Cin + cs * 8) * 2);
    }
    const int slices = Cin / 64;
    int la_k = 0, la_slot = 0;          // next weight K-step to issue
    int lb_r = 0, lb_c = 0, lb_slot = 0;
    auto dma_a_part = [&](int i0, int i1) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < G::WLW; ++i)
            if (i >= i0 && i < i1)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_char*)(size_t)(lds0 + la_slot * G::ASZ + (8 * i + wave) * 1024), 16, woff[i], (la_k % 36) * 128, 0, 0);
    };
    auto advance_a = [&]() __attribute__((always_inline)) {
        la_slot = la_slot == 2 ? 0 : la_slot + 1;
        la_k = la_k + 1 == nk ? 0 : la_k + 1;
    };
    auto dma_a = [&]() __attribute__((always_inline)) { dma_a_part(0, G::WLW); advance_a(); };
    auto retile = [&](int it) __attribute__((always_inline)) {
        const int P0n = (int)(((long long)blockIdx.x + 256ll * it) * G::BP % (M - G::BP - 8));
#pragma unroll
        for (int j = 0; j < G::XLW; ++j) xoff[j] = (uint32_t)(((P0n + 8 * (8 * j + wave) + lrow) * Cin + cs * 8) * 2);
    };
    auto dma_b = [&](auto J0c, auto J1c) __attribute__((always_inline)) {
        constexpr int J0 = decltype(J0c)::value, J1 = decltype(J1c)::value;
        const uint32_t soff = (uint32_t)((lb_r * Wimg * Cin + lb_c * 64) * 2);
#pragma unroll
        for (int j = J0; j < J1; ++j) {
            const uint32_t dst = lds0 + (uint32_t)(8 * j + wave < G::NPB ? 3 * G::ASZ + lb_slot * G::BSZ + (8 * j + wave) * 1024 : G::DUMP);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_char*)(size_t)dst, 16, xoff[j], soff, 0, 0);
        }
    };
    auto advance_b = [&]() __attribute__((always_inline)) {
        lb_slot ^= 1;
        if (++lb_c == slices) { lb_c = 0; if (++lb_r == 3) lb_r = 0; }
    };
    typedef std::integral_constant<int, 0> C0;
    typedef std::integral_constant<int, (G::XLW + 1) / 2> CH;
    typedef std::integral_constant<int, G::XLW> CN;
    constexpr int NB0 = (G::XLW + 1) / 2, NB1 = G::XLW / 2;

    f32x4 acc[CBW][PBW];
#pragma unroll
    for (int i = 0; i < CBW; ++i)
#pragma unroll
        for (int j = 0; j < PBW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    s16x8 a[2][CBW], b[2][PBW];
    const uint32_t afrag = lds0 + (uint32_t)((wc * 16 * CBW + fr) * 128);
    const uint32_t brow0 = (uint32_t)(wp * 16 * PBW + fr);
    auto reads = [&](int set, int sa, int sb, auto Qc, int h) __attribute__((always_inline)) {
        constexpr int Q = decltype(Qc)::value;
        const uint32_t abase = afrag + (uint32_t)(sa * G::ASZ);
        const uint32_t brow = brow0 + Q;
        const uint32_t bbase = lds0 + (uint32_t)(3 * G::ASZ + sb * G::BSZ) + brow * 128u;
        const uint32_t zrow = lds0 + (uint32_t)G::ZOFF;
        const uint32_t kc = (uint32_t)(fq + 4 * h);
        const uint32_t ap = abase + ((kc ^ (uint32_t)(fr & 7)) << 4);
        const uint32_t bp = bbase + ((kc ^ (brow & 7u)) << 4);
        const uint32_t zsel = zrow + (bp & 2047u);
#pragma unroll
        for (int i = 0; i < CBW; ++i) a[set][i] = *reinterpret_cast<lds_fptr>((size_t)(ap + i * 2048));
#pragma unroll
        for (int j = 0; j < PBW; ++j) {
            uint32_t bj = bp;
            if constexpr (Q != 1) {
                const uint32_t t = (uint32_t)__builtin_amdgcn_sbfe((int)(Q == 0 ? hm0 : hm2), j, 1);
                bj = (t & (zsel - (uint32_t)(j * 2048))) | (~t & bp);
            }
            b[set][j] = *reinterpret_cast<lds_fptr>((size_t)(bj + j * 2048));
        }
    };
    auto mfmas = [&](int set) __attribute__((always_inline)) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < PBW; ++j)
#pragma unroll
            for (int i = 0; i < CBW; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[set][i], b[set][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    int pend_h = 0;
    auto lgkm0 = [&]() __attribute__((always_inline)) { __builtin_amdgcn_s_waitcnt(0xC07F); };

    // MODE 1: the interval's pieces between the MFMAs (after every CBW * 2 of them)
    auto mfmas_with = [&](auto Qc, int h) __attribute__((always_inline)) {
        constexpr int Q = decltype(Qc)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < PBW; ++j) {
#pragma unroll
            for (int i = 0; i < CBW; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][i], b[0][j], acc[i][j], 0, 0, 0);
            if (j == 1) {
                if (h == 0) {
                    if constexpr (Q == 0) dma_b(C0{}, CH{});
                    if constexpr (Q == 1) { dma_b(CH{}, CN{}); advance_b(); }
                } else dma_a_part(0, 1);
            }
            if (j == 3) { if (h == 0) dma_a_part(0, Q == 2 ? G::WLW / 2 : (G::WLW + 3) / 4); else dma_a_part(1, 2); }
            if (j == 5 && h == 1) { dma_a_part(2 > (Q == 2 ? G::WLW / 2 : (G::WLW + 3) / 4) ? 2 : (Q == 2 ? G::WLW / 2 : (G::WLW + 3) / 4), G::WLW); advance_a(); }
        }
        __builtin_amdgcn_s_setprio(0);
    };
    // prologue: K-steps 0, 1 (weights), group 0 (activations)
    dma_b(C0{}, CN{}); advance_b();
    dma_a(); dma_a();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    sync();

    int sa = 0, sb = 0;
    constexpr bool g1 = GRP == 1;
    // pieces of one K-step: part 0 = activation share (Q = 0, 1) + first weight pieces, part 1 = the other weight pieces (3 + 3, 3 + 3, 2 + 2)
    auto part = [&](auto Qc, int h) __attribute__((always_inline)) {
        constexpr int Q = decltype(Qc)::value;
        if constexpr (MODE == 5) return;
        if (h == 0) {
            if constexpr (Q == 0) dma_b(C0{}, CH{});
            if constexpr (Q == 1) { dma_b(CH{}, CN{}); advance_b(); }
            dma_a_part(0, Q == 2 ? G::WLW / 2 : (G::WLW + 3) / 4);
        } else {
            dma_a_part(Q == 2 ? G::WLW / 2 : (G::WLW + 3) / 4, G::WLW);
            advance_a();
        }
    };
    auto waitv = [&](auto Qc) __attribute__((always_inline)) {
        constexpr int Q = decltype(Qc)::value;
        if constexpr (MODE == 5) return;
        if constexpr (Q == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::WLW + NB0) : "memory");
        else if constexpr (Q == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::WLW + NB1) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::WLW) : "memory");
    };
    auto rd = [&](auto Qc, int h) __attribute__((always_inline)) {
        if constexpr (MODE == 4) part(Qc, h);
        reads(0, sa, sb, Qc, h);
        if constexpr (MODE == 0 || MODE == 5) part(Qc, h);
        lgkm0();
    };
    auto mm = [&](auto Qc, int h) __attribute__((always_inline)) {
        if constexpr (MODE == 2) part(Qc, h);
        if constexpr (MODE == 1) mfmas_with(Qc, h); else mfmas(0);
        if constexpr (MODE == 3) part(Qc, h);
    };
    auto kstep = [&](auto Qc) __attribute__((always_inline)) {
        if constexpr (!g1) {
            rd(Qc, 0);
            sync();
            mm(Qc, 0);
            sync();
            rd(Qc, 1);
            sync();
            mm(Qc, 1);
            waitv(Qc);
            sync();
        } else {
            mm(Qc, 1);            // (the pieces of the previous K-step's second part: same counts)
            sync();
            rd(Qc, 0);
            sync();
            mm(Qc, 0);
            sync();
            rd(Qc, 1);
            waitv(Qc);
            sync();
        }
        sa = sa == 2 ? 0 : sa + 1;
        if constexpr (decltype(Qc)::value == 2) sb ^= 1;
    };
    reads(0, 0, 0, C0{}, 0);
    lgkm0();
    for (int s = 0; s < nk; s += 3) {
        if (s % 36 == 0 && s > 0) retile(s / 36);
        kstep(std::integral_constant<int, 0>{});
        kstep(std::integral_constant<int, 1>{});
        kstep(std::integral_constant<int, 2>{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CBW; ++i)
#pragma unroll
        for (int j = 0; j < PBW; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * 512 + tid] = s;
}

template <int MODE, int WC, int WP, int CBW, int PBW>
__global__ __launch_bounds__(512, 2) void k(const char* __restrict__ Wt, const char* __restrict__ X, float* __restrict__ out, int nk, int Kpad,
                                            int Cin, int Wimg, int M, uint32_t hm0, uint32_t hm2) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave < 4) body<MODE, 0, WC, WP, CBW, PBW>(smem, wave, Wt, X, out, nk, Kpad, Cin, Wimg, M, hm0, hm2);
    else body<MODE, 1, WC, WP, CBW, PBW>(smem, wave, Wt, X, out, nk, Kpad, Cin, Wimg, M, hm0, hm2);
#endif
}

template <int MODE, int WC, int WP, int CBW, int PBW> static void run(const char* name, const char* W, const char* X, float* out, int Cin, int Wimg, int M) {
    typedef Cfg<WC, WP, CBW, PBW> G;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE, WC, WP, CBW, PBW>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int Kpad = 9 * Cin, nk = 9 * Cin / 64 * 40;      // 40 tiles' worth of K-steps
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, WC, WP, CBW, PBW>), dim3(256), dim3(512), G::LDS, 0, W, X, out, nk, Kpad, Cin, Wimg, M, 0x01u, 0x40u);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    if (hipGetLastError() != hipSuccess) { printf("%s: launch failed\n", name); return; }
    const double flop = 256.0 * nk * 2.0 * G::BM * G::BP * 64;
    printf("%-40s tile %3d x %3d  LDS %6d  %7.2f ms  %6.0f TFLOP/s\n", name, G::BM, G::BP, G::LDS, best, flop / best / 1e9);
}

int main() {
    const int Cin = 256, Wimg = 14, M = 256 * 196;
    std::vector<short> h((size_t)M * Cin);
    for (auto& v : h) v = (short)(0x3f80 + (rand() & 0x7f) - ((rand() & 1) ? 0x8000 : 0));     // bf16 +-[1, 2)
    char *W, *X; float* out;
    if (hipMalloc(&W, 256 * 9 * Cin * 2) != hipSuccess || hipMalloc(&X, (size_t)M * Cin * 2) != hipSuccess || hipMalloc(&out, 256 * 512 * 4) != hipSuccess) return 1;
    (void)hipMemcpy(X, h.data(), (size_t)M * Cin * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(W, h.data(), 256 * 9 * Cin * 2, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 4, 2, 4, 7>("pieces behind the reads", W, X, out, Cin, Wimg, M);
        run<4, 4, 2, 4, 7>("pieces in front of the reads", W, X, out, Cin, Wimg, M);
        run<1, 4, 2, 4, 7>("pieces between the MFMAs", W, X, out, Cin, Wimg, M);
        run<2, 4, 2, 4, 7>("pieces in front of the MFMAs", W, X, out, Cin, Wimg, M);
        run<3, 4, 2, 4, 7>("pieces behind the MFMAs", W, X, out, Cin, Wimg, M);
        run<5, 4, 2, 4, 7>("no pieces", W, X, out, Cin, Wimg, M);
    }
    return 0;
}

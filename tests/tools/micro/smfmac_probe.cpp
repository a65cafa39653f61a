// Dev tool (GPU box): operand layout and cost of v_smfmac_f32_16x16x64_f16 (2:4 structured-sparse A, gfx950), probed empirically -
// there is no ISA document on the box. hipcc -O3 --offload-arch=gfx950 smfmac_probe.cpp -o /tmp/smfmac_probe && /tmp/smfmac_probe
//   A (compressed, 8 halfs per lane): lane (row i = l % 16, q = l / 16) slot s holds the value 1 + 8 q + s (same for every row)
//   idx: every 4-bit field = p0 | p1 << 2 (positions of a group's two kept elements), all fields equal
//   B (16 halfs per lane): one-hot - element e of lane (column 0, quarter kq)
//   D[:, 0] then names the compressed slot that the dense K index of B's element meets (0: none).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) _Float16 f16x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ void probe(float* out) {          // out[trial][16 rows]
    const int l = threadIdx.x, q = l >> 4;
    f16x8 a;
    for (int s = 0; s < 8; ++s) a[s] = (_Float16)(float)(1 + 8 * q + s);
    const int pats[6][2] = {{0, 1}, {0, 2}, {0, 3}, {1, 2}, {1, 3}, {2, 3}};
    int trial = 0;
    for (int kq = 0; kq < 4; ++kq)
        for (int e = 0; e < 16; ++e)
            for (int pt = 0; pt < 6; ++pt, ++trial) {
                f16x16 b;
                for (int i = 0; i < 16; ++i) b[i] = (_Float16)((l == 16 * kq && i == e) ? 1.f : 0.f);
                const int f = pats[pt][0] | (pats[pt][1] << 2);
                int idx = 0;
                for (int g = 0; g < 8; ++g) idx |= f << (4 * g);
                f32x4 c = {0.f, 0.f, 0.f, 0.f};
                c = __builtin_amdgcn_smfmac_f32_16x16x64_f16(a, b, c, idx, 0, 0);
                if ((l & 15) == 0)
                    for (int r = 0; r < 4; ++r) out[trial * 16 + 4 * q + r] = c[r];
            }
}

// cost: N dependent-free instructions per iteration on 4 accumulators, one wave per SIMD (256 threads), s_memtime around the loop
template <int SPARSE> __global__ void cost(const f16x8* ap, const f16x16* bp, float* sink, unsigned long long* cyc) {
    f16x8 a = ap[threadIdx.x & 63];
    f16x16 b = bp[threadIdx.x & 63];
    f16x8 b8;
    for (int i = 0; i < 8; ++i) b8[i] = b[i];
    f32x4 c[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 2000; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (SPARSE == 1) c[j] = __builtin_amdgcn_smfmac_f32_16x16x64_f16(a, b, c[j], 0x44444444, 0, 0);
            else c[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b8, c[j], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int j = 0; j < 4; ++j) s += c[j][0] + c[j][1] + c[j][2] + c[j][3];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    const int trials = 4 * 16 * 6;
    float* d;
    hipMalloc(&d, trials * 16 * sizeof(float));
    hipMemset(d, 0, trials * 16 * sizeof(float));
    probe<<<1, 64>>>(d);
    std::vector<float> h(trials * 16);
    hipMemcpy(h.data(), d, h.size() * sizeof(float), hipMemcpyDeviceToHost);
    const char* pn[6] = {"01", "02", "03", "12", "13", "23"};
    for (int kq = 0; kq < 4; ++kq)
        for (int e = 0; e < 16; ++e) {
            printf("B lane-quarter %d element %2d:", kq, e);
            for (int pt = 0; pt < 6; ++pt) {
                const float* r = &h[((kq * 16 + e) * 6 + pt) * 16];
                bool same = true;
                for (int i = 1; i < 16; ++i) same &= r[i] == r[0];
                printf("  idx%s->%s%g", pn[pt], same ? "" : "ROWS-DIFFER ", r[0]);
            }
            printf("\n");
        }
    // cost
    f16x8* ap; f16x16* bp; float* sink; unsigned long long* cyc;
    hipMalloc(&ap, 64 * sizeof(f16x8)); hipMalloc(&bp, 64 * sizeof(f16x16)); hipMalloc(&sink, 256 * 256 * sizeof(float)); hipMalloc(&cyc, 256 * 8);
    hipMemset(ap, 0x3c, 64 * sizeof(f16x8)); hipMemset(bp, 0x3c, 64 * sizeof(f16x16));
    for (int rep = 0; rep < 2; ++rep) {
        unsigned long long hc[2] = {0, 0};
        cost<0><<<256, 256>>>(ap, bp, sink, cyc); hipMemcpy(&hc[0], cyc, 8, hipMemcpyDeviceToHost);
        cost<1><<<256, 256>>>(ap, bp, sink, cyc); hipMemcpy(&hc[1], cyc, 8, hipMemcpyDeviceToHost);
        printf("cycles per instruction (one wave per SIMD, 4 accumulators): dense 16x16x32 %.2f   sparse 16x16x64 %.2f\n", hc[0] / 8000.0, hc[1] / 8000.0);
    }
    return 0;
}

/*
 * ORACLE - test infrastructure, not product code.
 *
 * ATen-free plain-C restatement of the individual ops on the reference's conv-net inference path,
 * used to cross-check oracle/refnet.py (which delegates to torch like the reference does) and as the
 * per-op checker in tests/. NCHW fp32 tensors, double accumulation, torch semantics:
 * cross-correlation, zero padding, floor output size, BN eps inside the sqrt, max-pool pads with -inf.
 *
 * Reference lines restated (relative to the reference root):
 *   conv2d        pytorchcv/models/common/conv.py:250-258 (nn.Conv2d incl. groups/dilation), :245-249 (4-tuple pad)
 *   bn_eval       pytorchcv/models/common/norm.py:34-50   (nn.BatchNorm2d, eval)
 *   act           pytorchcv/models/common/activ.py:16-81,117-132 (Swish, HSigmoid, HSwish, ReLU, ReLU6, Sigmoid)
 *   maxpool2d     pytorchcv/models/resnet.py:255-258      (MaxPool2d(3, 2, 1))
 *   avgpool2d     pytorchcv/models/resnet.py:316-318      (AvgPool2d(7, 1))
 *   linear        pytorchcv/models/resnet.py:320-322
 *   se_gate       pytorchcv/models/common/att.py:94-105
 *
 * Parity pin: tests/test_oracle_golden.py checks these against the golden block fixtures generated
 * from the imported reference (tests/golden/make_golden.py).
 */
#include <math.h>
#include <stddef.h>

static long out_size(long in, long k, long s, long p0, long p1, long d) {
    return (in + p0 + p1 - d * (k - 1) - 1) / s + 1;
}

/* x[N,C,H,W], w[O,C/g,kh,kw], bias[O] or NULL -> y[N,O,Ho,Wo] */
int cref_conv2d(const float* x, const float* w, const float* bias, float* y,
                int N, int C, int H, int W, int O, int kh, int kw, int sh, int sw,
                int pt, int pl, int pb, int pr, int dh, int dw, int groups) {
    if (C % groups || O % groups) return -1;
    const int Cg = C / groups, Og = O / groups;
    const long Ho = out_size(H, kh, sh, pt, pb, dh), Wo = out_size(W, kw, sw, pl, pr, dw);
    for (int n = 0; n < N; ++n)
        for (int o = 0; o < O; ++o) {
            const int g = o / Og;
            for (long ho = 0; ho < Ho; ++ho)
                for (long wo = 0; wo < Wo; ++wo) {
                    double acc = bias ? (double)bias[o] : 0.0;
                    for (int c = 0; c < Cg; ++c)
                        for (int r = 0; r < kh; ++r) {
                            const long hi = ho * sh - pt + (long)r * dh;
                            if (hi < 0 || hi >= H) continue;
                            for (int q = 0; q < kw; ++q) {
                                const long wi = wo * sw - pl + (long)q * dw;
                                if (wi < 0 || wi >= W) continue;
                                acc += (double)x[(((size_t)n * C + g * Cg + c) * H + hi) * W + wi] *
                                       (double)w[(((size_t)o * Cg + c) * kh + r) * kw + q];
                            }
                        }
                    y[(((size_t)n * O + o) * Ho + ho) * Wo + wo] = (float)acc;
                }
        }
    return 0;
}

/* in place: y = (x - mean) / sqrt(var + eps) * gamma + beta */
int cref_bn_eval(float* x, const float* gamma, const float* beta, const float* mean, const float* var,
                 float eps, int N, int C, long HW) {
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c) {
            const double inv = 1.0 / sqrt((double)var[c] + (double)eps);
            float* p = x + ((size_t)n * C + c) * HW;
            for (long i = 0; i < HW; ++i)
                p[i] = (float)(((double)p[i] - (double)mean[c]) * inv * (double)gamma[c] + (double)beta[c]);
        }
    return 0;
}

/* in place; act: 0 none, 1 relu, 2 relu6, 3 sigmoid, 4 swish, 5 hsigmoid, 6 hswish */
int cref_act(float* x, long n, int act) {
    for (long i = 0; i < n; ++i) {
        float v = x[i];
        if (act == 1) v = v > 0.f ? v : 0.f;
        else if (act == 2) v = v < 0.f ? 0.f : (v > 6.f ? 6.f : v);
        else if (act == 3) v = (float)(1.0 / (1.0 + exp(-(double)v)));
        else if (act == 4) v = v * (float)(1.0 / (1.0 + exp(-(double)v)));
        else if (act == 5) { float t = v + 3.f; t = t < 0.f ? 0.f : (t > 6.f ? 6.f : t); v = t / 6.f; }
        else if (act == 6) { float t = v + 3.f; t = t < 0.f ? 0.f : (t > 6.f ? 6.f : t); v = v * t / 6.f; }
        x[i] = v;
    }
    return 0;
}

int cref_add(float* x, const float* r, long n) {
    for (long i = 0; i < n; ++i) x[i] = x[i] + r[i];
    return 0;
}

int cref_maxpool2d(const float* x, float* y, int N, int C, int H, int W, int k, int s, int p) {
    const long Ho = out_size(H, k, s, p, p, 1), Wo = out_size(W, k, s, p, p, 1);
    for (long nc = 0; nc < (long)N * C; ++nc)
        for (long ho = 0; ho < Ho; ++ho)
            for (long wo = 0; wo < Wo; ++wo) {
                float m = -INFINITY;
                for (int r = 0; r < k; ++r)
                    for (int q = 0; q < k; ++q) {
                        const long hi = ho * s - p + r, wi = wo * s - p + q;
                        if (hi < 0 || hi >= H || wi < 0 || wi >= W) continue;
                        const float v = x[((size_t)nc * H + hi) * W + wi];
                        if (v > m) m = v;
                    }
                y[((size_t)nc * Ho + ho) * Wo + wo] = m;
            }
    return 0;
}

/* no padding, as AvgPool2d(k, stride=s) */
int cref_avgpool2d(const float* x, float* y, int N, int C, int H, int W, int k, int s) {
    const long Ho = out_size(H, k, s, 0, 0, 1), Wo = out_size(W, k, s, 0, 0, 1);
    for (long nc = 0; nc < (long)N * C; ++nc)
        for (long ho = 0; ho < Ho; ++ho)
            for (long wo = 0; wo < Wo; ++wo) {
                double a = 0.0;
                for (int r = 0; r < k; ++r)
                    for (int q = 0; q < k; ++q)
                        a += (double)x[((size_t)nc * H + ho * s + r) * W + wo * s + q];
                y[((size_t)nc * Ho + ho) * Wo + wo] = (float)(a / (double)(k * k));
            }
    return 0;
}

/* y[N,O] = x[N,K] w[O,K]^T + b */
int cref_linear(const float* x, const float* w, const float* b, float* y, int N, int K, int O) {
    for (int n = 0; n < N; ++n)
        for (int o = 0; o < O; ++o) {
            double a = b ? (double)b[o] : 0.0;
            for (int k = 0; k < K; ++k) a += (double)x[(size_t)n * K + k] * (double)w[(size_t)o * K + k];
            y[(size_t)n * O + o] = (float)a;
        }
    return 0;
}

/* SE gate: g[N,C] = sigmoid(W2 relu(W1 mean_hw(x) + b1) + b2); W1[M,C], W2[C,M] */
int cref_se_gate(const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                 float* g, float* tmp_mean, float* tmp_mid, int N, int C, long HW, int M) {
    for (int n = 0; n < N; ++n) {
        for (int c = 0; c < C; ++c) {
            double a = 0.0;
            const float* p = x + ((size_t)n * C + c) * HW;
            for (long i = 0; i < HW; ++i) a += (double)p[i];
            tmp_mean[c] = (float)(a / (double)HW);
        }
        for (int m = 0; m < M; ++m) {
            double a = (double)b1[m];
            for (int c = 0; c < C; ++c) a += (double)w1[(size_t)m * C + c] * (double)tmp_mean[c];
            tmp_mid[m] = a > 0.0 ? (float)a : 0.f;
        }
        for (int c = 0; c < C; ++c) {
            double a = (double)b2[c];
            for (int m = 0; m < M; ++m) a += (double)w2[(size_t)c * M + m] * (double)tmp_mid[m];
            g[(size_t)n * C + c] = (float)(1.0 / (1.0 + exp(-a)));
        }
    }
    return 0;
}

// d3q_inst.hpp - the instantiations of d3q_kernel (d3q_bf16.hip / d3q_f16.hip define them, pcv_api.hip sees `extern template`).
//   X(DT, WC, WP, CBW, PBW, KS): block tile = (16 CBW WC) channels x (16 PBW WP) pixels, KS K-halves per section
#pragma once
#include "d3q_conv.hpp"

#define D3Q_SHAPES(X, DT)         \
    X(DT, 8, 1, 2, 7, 2)          /* 0: 256 ch x 112 px, wave 32 x 112 */  \
    X(DT, 4, 2, 2, 7, 2)          /* 1: 128 x 224 */                       \
    X(DT, 2, 4, 2, 7, 2)          /* 2: 64 x 448 */                        \
    X(DT, 8, 1, 2, 4, 2)          /* 3: 256 x 64 */                        \
    X(DT, 4, 2, 2, 4, 2)          /* 4: 128 x 128 */                       \
    X(DT, 2, 4, 2, 4, 2)          /* 5: 64 x 256 */                        \
    X(DT, 8, 1, 2, 7, 1)          /* 6: 256 x 112, K-half sections (A/B of the section length) */ \
    X(DT, 2, 4, 2, 7, 1)          /* 7: 64 x 448, K-half sections */

// 1x1 / stride 1 mode (K >= 512): X(DT, WC, WP, CBW, PBW, KS)
#define D3Q1_SHAPES(X, DT)        \
    X(DT, 4, 2, 2, 7, 2)          /* 0: 128 ch x 224 px */ \
    X(DT, 8, 1, 2, 7, 2)          /* 1: 256 x 112 */
#define D3Q1_DEFINE(DT, WC, WP, CBW, PBW, KS) template __global__ void d3q_kernel<DT, WC, WP, CBW, PBW, KS, true>(const D3Params);
#define D3Q1_DECLARE(DT, WC, WP, CBW, PBW, KS) extern template __global__ void d3q_kernel<DT, WC, WP, CBW, PBW, KS, true>(const D3Params);

#define D3Q_DEFINE(DT, WC, WP, CBW, PBW, KS) template __global__ void d3q_kernel<DT, WC, WP, CBW, PBW, KS>(const D3Params);
#define D3Q_DECLARE(DT, WC, WP, CBW, PBW, KS) extern template __global__ void d3q_kernel<DT, WC, WP, CBW, PBW, KS>(const D3Params);

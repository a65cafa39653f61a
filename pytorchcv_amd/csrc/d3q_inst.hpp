// d3q_inst.hpp - the instantiations of d3q_kernel (d3q_bf16.hip / d3q_f16.hip define them, pcv_api.hip sees `extern template`).
//   X(DT, WC, WP, CBW, PBW, KS): block tile = (16 CBW WC) channels x (16 PBW WP) pixels, KS K-halves per section
#pragma once
#include "d3q_conv.hpp"

#define D3Q_SHAPES(X, DT)         \
    X(DT, 8, 1, 2, 13, 1)         /* 0: 256 ch x 208 px, wave 32 x 208 */  \
    X(DT, 4, 2, 4, 7, 2)          /* 1: 256 x 224, wave 64 x 112 */        \
    X(DT, 8, 1, 2, 7, 2)          /* 2: 256 x 112, wave 32 x 112 */        \
    X(DT, 4, 2, 2, 13, 1)         /* 3: 128 x 416, wave 32 x 208 */        \
    X(DT, 4, 2, 2, 7, 2)          /* 4: 128 x 224, wave 32 x 112 */        \
    X(DT, 2, 4, 2, 7, 2)          /* 5: 64 x 448, wave 32 x 112 */         \
    X(DT, 2, 4, 2, 4, 2)          /* 6: 64 x 256, wave 32 x 64 */          \
    X(DT, 4, 2, 4, 7, 1)          /* 7: 256 x 224, K-half sections */      \
    X(DT, 8, 1, 2, 7, 1)          /* 8: 256 x 112, K-half sections */

#define D3Q_DEFINE(DT, WC, WP, CBW, PBW, KS) template __global__ void d3q_kernel<DT, WC, WP, CBW, PBW, KS>(const D3Params);
#define D3Q_DECLARE(DT, WC, WP, CBW, PBW, KS) extern template __global__ void d3q_kernel<DT, WC, WP, CBW, PBW, KS>(const D3Params);

// d3x3_inst.hpp - the instantiations of d3x3_kernel (d3x3_bf16.hip / d3x3_f16.hip define them, pcv_api.hip sees `extern template`).
//   X(DT, WC, WP, CBW, PBW, KS): block tile = (16 CBW WC) channels x (16 PBW WP) pixels, KS K-halves per section
#pragma once
#include "d3x3_conv.hpp"

#define D3X3_SHAPES(X, DT)        \
    X(DT, 2, 4, 8, 4, 1)          /* 0: 256 ch x 256 px, wave 128 x 64 */  \
    X(DT, 4, 2, 4, 7, 2)          /* 1: 256 x 224, wave 64 x 112 */        \
    X(DT, 8, 1, 2, 13, 1)         /* 2: 256 x 208, wave 32 x 208 */        \
    X(DT, 8, 1, 2, 7, 2)          /* 3: 256 x 112, wave 32 x 112 */        \
    X(DT, 4, 2, 2, 13, 1)         /* 4: 128 x 416, wave 32 x 208 */        \
    X(DT, 2, 4, 4, 7, 2)          /* 5: 128 x 448, wave 64 x 112 */        \
    X(DT, 4, 2, 2, 7, 2)          /* 6: 128 x 224, wave 32 x 112 */        \
    X(DT, 2, 4, 2, 7, 2)          /* 7: 64 x 448, wave 32 x 112 */         \
    X(DT, 2, 4, 2, 4, 2)          /* 8: 64 x 256, wave 32 x 64 */          \
    X(DT, 8, 1, 2, 7, 1)          /* 9: 256 x 112 with K-half sections (A/B of the section length) */ \
    X(DT, 4, 2, 4, 7, 1)          /* 10: 256 x 224, K-half sections */ \
    X(DT, 2, 4, 4, 7, 1)          /* 11: 128 x 448, K-half sections */

#define D3X3_DEFINE(DT, WC, WP, CBW, PBW, KS) template __global__ void d3x3_kernel<DT, WC, WP, CBW, PBW, KS>(const D3Params);
#define D3X3_DECLARE(DT, WC, WP, CBW, PBW, KS) extern template __global__ void d3x3_kernel<DT, WC, WP, CBW, PBW, KS>(const D3Params);

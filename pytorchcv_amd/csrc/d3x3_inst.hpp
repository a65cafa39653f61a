// d3x3_inst.hpp - the instantiations of d3x3_kernel (d3x3_bf16.hip / d3x3_f16.hip define them, pcv_api.hip sees `extern template`).
//   X(DT, WC, WP, CBW, PBW): block tile = (16 CBW WC) channels x (16 PBW WP) pixels
#pragma once
#include "d3x3_conv.hpp"

#define D3X3_SHAPES(X, DT)     \
    X(DT, 2, 4, 8, 4)          /* 0: 256 ch x 256 px, wave 128 x 64 */  \
    X(DT, 4, 2, 4, 7)          /* 1: 256 x 224, wave 64 x 112 */        \
    X(DT, 8, 1, 2, 13)         /* 2: 256 x 208, wave 32 x 208 */        \
    X(DT, 8, 1, 2, 7)          /* 3: 256 x 112, wave 32 x 112 */        \
    X(DT, 4, 2, 2, 13)         /* 4: 128 x 416 */                       \
    X(DT, 4, 2, 2, 7)          /* 5: 128 x 224 */                       \
    X(DT, 2, 4, 2, 7)          /* 6: 64 x 448 */                        \
    X(DT, 2, 4, 2, 4)          /* 7: 64 x 256 */

#define D3X3_DEFINE(DT, WC, WP, CBW, PBW) template __global__ void d3x3_kernel<DT, WC, WP, CBW, PBW>(const D3Params);
#define D3X3_DECLARE(DT, WC, WP, CBW, PBW) extern template __global__ void d3x3_kernel<DT, WC, WP, CBW, PBW>(const D3Params);

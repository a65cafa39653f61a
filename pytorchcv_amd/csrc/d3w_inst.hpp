// d3w_inst.hpp - the instantiations of d3w_kernel (d3w_bf16.hip / d3w_f16.hip define them, pcv_api.hip sees `extern template`).
//   X(DT, WC, WP, CBW, PBW): block tile = (16 CBW WC) channels x (16 PBW WP) pixels, eight self-loading waves
#pragma once
#include "d3w_conv.hpp"

#define D3W_SHAPES(X, DT)      \
    X(DT, 4, 2, 4, 7)          /* 0: 256 ch x 224 px, wave 64 x 112 */  \
    X(DT, 4, 2, 2, 13)         /* 1: 128 x 416, wave 32 x 208 */        \
    X(DT, 2, 4, 4, 6)          /* 2: 128 x 384, wave 64 x 96 */         \
    X(DT, 4, 2, 2, 7)          /* 3: 128 x 224, wave 32 x 112 */        \
    X(DT, 2, 4, 2, 7)          /* 4: 64 x 448, wave 32 x 112 */

#define D3W_DEFINE(DT, WC, WP, CBW, PBW) template __global__ void d3w_kernel<DT, WC, WP, CBW, PBW>(const D3Params);
#define D3W_DECLARE(DT, WC, WP, CBW, PBW) extern template __global__ void d3w_kernel<DT, WC, WP, CBW, PBW>(const D3Params);

// d3w_inst.hpp - the instantiations of d3w_kernel (d3w_bf16.hip / d3w_f16.hip define them, pcv_api.hip sees `extern template`).
//   X(DT, WC, WP, CBW, PBW, KS, NSA): block tile = (16 CBW WC) channels x (16 PBW WP) pixels, eight self-loading waves, KS K-halves per interval, NSA weight-ring slots
#pragma once
#include "d3w_conv.hpp"

#define D3W_SHAPES(X, DT)      \
    X(DT, 4, 2, 4, 7, 1, 3)     /* 0: 256 ch x 224 px, wave 64 x 112, K-half intervals */      \
    X(DT, 4, 2, 4, 7, 2, 3)     /* 1: 256 x 224, whole K-steps per interval */                 \
    X(DT, 2, 4, 4, 7, 1, 2)     /* 2: 128 x 448, wave 64 x 112, two-slot weight ring */        \
    X(DT, 2, 4, 2, 7, 2, 3)     /* 3: 64 x 448, wave 32 x 112, whole K-steps per interval */

// shapes with trimmed tiles: X(DT, WC, WP, CBW, PBW, KS, NSA, TRIM)
#define D3WT_SHAPES(X, DT)     \
    X(DT, 2, 4, 4, 7, 1, 3, 2)  /* 4: 128 x 416 (4 x 7 blocks less 2), three-slot weight ring */
#define D3WT_DEFINE(DT, WC, WP, CBW, PBW, KS, NSA, TRIM) template __global__ void d3w_kernel<DT, WC, WP, CBW, PBW, KS, NSA, TRIM>(const D3Params);
#define D3WT_DECLARE(DT, WC, WP, CBW, PBW, KS, NSA, TRIM) extern template __global__ void d3w_kernel<DT, WC, WP, CBW, PBW, KS, NSA, TRIM>(const D3Params);

#define D3W_DEFINE(DT, WC, WP, CBW, PBW, KS, NSA) template __global__ void d3w_kernel<DT, WC, WP, CBW, PBW, KS, NSA>(const D3Params);
#define D3W_DECLARE(DT, WC, WP, CBW, PBW, KS, NSA) extern template __global__ void d3w_kernel<DT, WC, WP, CBW, PBW, KS, NSA>(const D3Params);

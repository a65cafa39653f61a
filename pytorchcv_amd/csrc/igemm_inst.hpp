// igemm_inst.hpp - the list of igemm_conv_kernel instantiations, one translation unit per storage type
// (igemm_bf16.hip / igemm_f16.hip / igemm_f32.hip define them, pcv_api.hip sees `extern template`).
//   X(DT, OT, CB, PB, WC, WP, RAGGED, KHW)
#pragma once
#include "igemm_conv.hpp"

#define IGEMM_TILES(X, DT, KHW)                     \
    X(DT, DT, 2, 4, 1, 4, false, KHW)               \
    X(DT, DT, 4, 4, 1, 4, false, KHW)               \
    X(DT, DT, 4, 4, 2, 2, false, KHW)               \
    X(DT, DT, 4, 4, 4, 1, false, KHW)

// regular kernels for generic / 1x1 / 3x3 taps, plus the ragged-channel and fp32-logit variants (128x128 tile only)
// half-height pixel tiles for the HBM-bound 1x1 layers: 48 KB of LDS and ~half the accumulators -> 3 blocks per CU
#define IGEMM_TILES_1X1_SMALL(X, DT)                \
    X(DT, DT, 4, 2, 1, 4, false, 1)                 \
    X(DT, DT, 4, 2, 2, 2, false, 1)

#define IGEMM_INSTANCES_SAMETYPE(X, DT)             \
    IGEMM_TILES(X, DT, 0)                           \
    IGEMM_TILES(X, DT, 1)                           \
    IGEMM_TILES_1X1_SMALL(X, DT)                    \
    IGEMM_TILES(X, DT, 9)                           \
    X(DT, DT, 4, 4, 2, 2, true, 0)
#define IGEMM_INSTANCES(X, DT)                      \
    IGEMM_INSTANCES_SAMETYPE(X, DT)                 \
    X(DT, PCV_F32, 4, 4, 2, 2, true, 0)

#define IGEMM_DEFINE(DT, OT, CB, PB, WC, WP, RG, KHW) \
    template __global__ void igemm_conv_kernel<DT, OT, CB, PB, WC, WP, RG, KHW>(const IgemmParams);
#define IGEMM_DECLARE(DT, OT, CB, PB, WC, WP, RG, KHW) \
    extern template __global__ void igemm_conv_kernel<DT, OT, CB, PB, WC, WP, RG, KHW>(const IgemmParams);

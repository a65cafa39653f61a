// igemm_inst.hpp - the list of igemm_conv_kernel instantiations, one translation unit per storage type
// (igemm_bf16.hip / igemm_f16.hip / igemm_f32.hip define them, pcv_api.hip sees `extern template`).
#pragma once
#include "igemm_conv.hpp"

#define IGEMM_INSTANCES(X, DT)                 \
    X(DT, DT, 2, 4, 1, 4, false)               \
    X(DT, DT, 4, 4, 1, 4, false)               \
    X(DT, DT, 4, 4, 2, 2, false)               \
    X(DT, DT, 4, 4, 4, 1, false)               \
    X(DT, DT, 4, 4, 2, 2, true)                \
    X(DT, PCV_F32, 4, 4, 2, 2, true)

#define IGEMM_DEFINE(DT, OT, CB, PB, WC, WP, RG) \
    template __global__ void igemm_conv_kernel<DT, OT, CB, PB, WC, WP, RG>(const IgemmParams);
#define IGEMM_DECLARE(DT, OT, CB, PB, WC, WP, RG) \
    extern template __global__ void igemm_conv_kernel<DT, OT, CB, PB, WC, WP, RG>(const IgemmParams);

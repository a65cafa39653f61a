// hconv3x3_inst.hpp - instantiation list of hconv3x3_kernel (defined in hconv3x3.hip, declared `extern` for pcv_api.hip)
//   X(DT, WC, WP): 128 x 256 tile with 8 waves, 64 x 256 tile with 4 waves (64-channel layers)
#pragma once
#include "hconv3x3.hpp"

#define HCONV_CONFIGS(X, DT) \
    X(DT, 2, 4)              \
    X(DT, 1, 4)
#define HCONV_INSTANCES(X) HCONV_CONFIGS(X, PCV_BF16) HCONV_CONFIGS(X, PCV_F16) HCONV_CONFIGS(X, PCV_F32)
#define HCONV_DEFINE(DT, WC, WP) template __global__ void hconv3x3_kernel<DT, WC, WP>(const HConvParams);
#define HCONV_DECLARE(DT, WC, WP) extern template __global__ void hconv3x3_kernel<DT, WC, WP>(const HConvParams);

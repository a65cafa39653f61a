// conv3x3_inst.hpp - instantiation list of conv3x3_kernel: X(DT, WC, WP)
#pragma once
#include "conv3x3.hpp"
#define CONV3_INSTANCES(X, DT) \
    X(DT, 2, 4)                \
    X(DT, 1, 8)
#define CONV3_DEFINE(DT, WC, WP) template __global__ void conv3x3_kernel<DT, WC, WP>(const Conv3Params);
#define CONV3_DECLARE(DT, WC, WP) extern template __global__ void conv3x3_kernel<DT, WC, WP>(const Conv3Params);

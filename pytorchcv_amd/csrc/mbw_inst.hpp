// mbw_inst.hpp - the instantiations of mbw_kernel (mbw_bf16.hip / mbw_f16.hip define them, pcv_api.hip sees `extern template`).
//   X(DT, S, NRT, ACT, TW): stride, 16-row tiles of the projection, compile-time inner activation (-1: launch-time), pixel-block width
#pragma once
#include "mbw.hpp"

#define MBW_SHAPES3(X, DT, S, NRT) \
    X(DT, S, NRT, -1, 16) X(DT, S, NRT, -1, 8) X(DT, S, NRT, PCV_ACT_RELU, 16) X(DT, S, NRT, PCV_ACT_RELU, 8) \
    X(DT, S, NRT, PCV_ACT_RELU6, 16) X(DT, S, NRT, PCV_ACT_RELU6, 8)
#define MBW_SHAPES(X, DT) MBW_SHAPES3(X, DT, 1, 2) MBW_SHAPES3(X, DT, 1, 4) MBW_SHAPES3(X, DT, 2, 2) MBW_SHAPES3(X, DT, 2, 4)

// two expand K steps (Cin <= 64), weights from L2: stride 1 only, 64 / 96 projected channels
#define MBW2_SHAPES(X, DT) \
    X(DT, 1, 4, -1, 16, 2) X(DT, 1, 4, -1, 8, 2) X(DT, 1, 4, PCV_ACT_RELU, 16, 2) X(DT, 1, 4, PCV_ACT_RELU, 8, 2) \
    X(DT, 1, 4, PCV_ACT_RELU6, 16, 2) X(DT, 1, 4, PCV_ACT_RELU6, 8, 2)
// wide units: 96 projected channels (NRT = 6), two or three expand K steps, 2 pixel blocks per wave tile, stride 1, 1 x 16 blocks
#define MBW3_SHAPES(X, DT) \
    X(DT, 1, 6, -1, 16, 2, 2) X(DT, 1, 6, PCV_ACT_RELU, 16, 2, 2) X(DT, 1, 6, PCV_ACT_RELU6, 16, 2, 2) \
    X(DT, 1, 6, -1, 16, 3, 2) X(DT, 1, 6, PCV_ACT_RELU, 16, 3, 2) X(DT, 1, 6, PCV_ACT_RELU6, 16, 3, 2)
#define MBW3_DEFINE(DT, S, NRT, ACT, TW, KA, RB) template __global__ void mbw_kernel<DT, S, NRT, ACT, TW, KA, RB>(const MbParams);
#define MBW3_DECLARE(DT, S, NRT, ACT, TW, KA, RB) extern template __global__ void mbw_kernel<DT, S, NRT, ACT, TW, KA, RB>(const MbParams);
#define MBW2_DEFINE(DT, S, NRT, ACT, TW, KA) template __global__ void mbw_kernel<DT, S, NRT, ACT, TW, KA>(const MbParams);
#define MBW2_DECLARE(DT, S, NRT, ACT, TW, KA) extern template __global__ void mbw_kernel<DT, S, NRT, ACT, TW, KA>(const MbParams);

#define MBW_DEFINE(DT, S, NRT, ACT, TW) template __global__ void mbw_kernel<DT, S, NRT, ACT, TW>(const MbParams);
#define MBW_DECLARE(DT, S, NRT, ACT, TW) extern template __global__ void mbw_kernel<DT, S, NRT, ACT, TW>(const MbParams);

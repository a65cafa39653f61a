// mbr_inst.hpp - the instantiations of mbr_kernel (mbr_bf16.hip / mbr_f16.hip define them, pcv_api.hip sees `extern template`).
//   X(DT, NRT, ACT, RO): 16-row tiles of the projection, compile-time inner activation (-1: launch-time), output rows of a wave tile
#pragma once
#include "mbr.hpp"

//   S: stride (2: RO = 4 output rows from 9 window rows)
#define MBR_SHAPES2(X, DT, NRT, RO, S) X(DT, NRT, -1, RO, S) X(DT, NRT, PCV_ACT_RELU, RO, S) X(DT, NRT, PCV_ACT_RELU6, RO, S)
#define MBR_SHAPES(X, DT) MBR_SHAPES2(X, DT, 2, 7, 1) MBR_SHAPES2(X, DT, 4, 4, 1) MBR_SHAPES2(X, DT, 2, 4, 2) MBR_SHAPES2(X, DT, 4, 4, 2)
#define MBR_DEFINE(DT, NRT, ACT, RO, S) template __global__ void mbr_kernel<DT, NRT, ACT, RO, S>(const MbParams);
#define MBR_DECLARE(DT, NRT, ACT, RO, S) extern template __global__ void mbr_kernel<DT, NRT, ACT, RO, S>(const MbParams);

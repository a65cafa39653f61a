// mbr_inst.hpp - the instantiations of mbr_kernel (mbr_bf16.hip / mbr_f16.hip define them, pcv_api.hip sees `extern template`).
//   X(DT, NRT, ACT, RO, S, KA, AFL): 16-row tiles of the projection, compile-time inner activation (-1: launch-time), output rows of a wave
//   tile, stride (2: RO = 4 output rows from 9 window rows), K steps of the expand GEMM, depthwise fragments in LDS (1) or read from the
//   packed table (0: the 14x14 units with 64 input channels, whose 1x1 weights fill the LDS), waves per block (4 = one per SIMD, 512
//   registers: 96 -> 576 -> 96), expand weights in LDS (0: read from L2 per pass, as the fragments), x through LDS (1: single-chunk units)
#pragma once
#include "mbr.hpp"

// (No launch-time-activation instances: units with h-swish / swish / ... stay on mbw.hpp / mbconv.hpp. The fp16 instantiation with the
// activation read at run time computed wrong, run-to-run different rows on two shapes - only with all three range guards compiled in, and
// depending on unrelated code motion; the matrix-pipe distances in its code are the compiler's usual 8 wait states, bf16 and the
// compile-time activations are bit-exact on every test. Unexplained, so the fragile instances are not built: profiles/experiments/mbr_kernel.md.)
#define MBR_SHAPES2(X, DT, NRT, RO, S, KA, AFL, WV, WEL, XL) \
    X(DT, NRT, PCV_ACT_RELU, RO, S, KA, AFL, WV, WEL, XL) X(DT, NRT, PCV_ACT_RELU6, RO, S, KA, AFL, WV, WEL, XL)
#define MBR_SHAPES(X, DT)                                                                                      \
    MBR_SHAPES2(X, DT, 2, 7, 1, 1, true, 8, true, true) /* (first: preferred where its two x buffers per wave fit the LDS) */ \
    MBR_SHAPES2(X, DT, 2, 7, 1, 1, true, 8, true, false) MBR_SHAPES2(X, DT, 4, 4, 1, 1, true, 8, true, false) MBR_SHAPES2(X, DT, 2, 4, 2, 1, true, 8, true, false) \
    MBR_SHAPES2(X, DT, 4, 4, 2, 1, true, 8, true, false) MBR_SHAPES2(X, DT, 4, 4, 1, 2, false, 8, true, false) MBR_SHAPES2(X, DT, 6, 3, 1, 2, false, 8, true, false) \
    MBR_SHAPES2(X, DT, 6, 5, 1, 3, false, 4, false, false)
#define MBR_DEFINE(DT, NRT, ACT, RO, S, KA, AFL, WV, WEL, XL) template __global__ void mbr_kernel<DT, NRT, ACT, RO, S, KA, AFL, WV, WEL, XL>(const MbParams);
#define MBR_DECLARE(DT, NRT, ACT, RO, S, KA, AFL, WV, WEL, XL) extern template __global__ void mbr_kernel<DT, NRT, ACT, RO, S, KA, AFL, WV, WEL, XL>(const MbParams);

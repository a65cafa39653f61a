// igemm_f32.hip - PCV_F32 (exact-f32 MFMA) instantiations of the implicit-GEMM convolution kernel.
#include "igemm_inst.hpp"
#define IGEMM_INSTANCES_F32(X)                      \
    X(PCV_F32, PCV_F32, 2, 4, 1, 4, false)          \
    X(PCV_F32, PCV_F32, 4, 4, 1, 4, false)          \
    X(PCV_F32, PCV_F32, 4, 4, 2, 2, false)          \
    X(PCV_F32, PCV_F32, 4, 4, 4, 1, false)          \
    X(PCV_F32, PCV_F32, 4, 4, 2, 2, true)
IGEMM_INSTANCES_F32(IGEMM_DEFINE)

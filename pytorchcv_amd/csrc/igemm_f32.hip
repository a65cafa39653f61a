// igemm_f32.hip - PCV_F32 (exact-f32 MFMA) instantiations of the implicit-GEMM convolution kernel.
#include "igemm_inst.hpp"
IGEMM_INSTANCES_SAMETYPE(IGEMM_DEFINE, PCV_F32)

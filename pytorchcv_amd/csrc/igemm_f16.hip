// igemm_f16.hip - PCV_F16 instantiations of the implicit-GEMM convolution kernel.
#include "igemm_inst.hpp"
IGEMM_INSTANCES(IGEMM_DEFINE, PCV_F16)

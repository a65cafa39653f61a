// igemm_bf16.hip - PCV_BF16 instantiations of the implicit-GEMM convolution kernel.
#include "igemm_inst.hpp"
IGEMM_INSTANCES(IGEMM_DEFINE, PCV_BF16)
